"""TEST-ONLY: builds and loads tests/wave_emu/libppg_emu.so -- the kernel source of
predpreygrass_amd/csrc/ppg_kernel.h compiled with g++ against a lockstep wave emulator --
so the device code and the host wrappers can be exercised on a machine without a GPU.
The product package never imports this."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "wave_emu")
EMU_LIB = os.path.join(EMU_DIR, "libppg_emu.so")
_KERNEL_SOURCES = [
    os.path.join(EMU_DIR, "ppg_emu_part.cpp"), os.path.join(EMU_DIR, "wave_emu.h"),
    os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_kernel.h"),
    *[os.path.join(ROOT, "predpreygrass_amd", "csrc", f"ppg_env_{p}.h")   # struct Env's member functions, one file per phase
      for p in ("load", "move", "sort", "observe", "coop", "engage", "reproduce", "step")],
    os.path.join(ROOT, "include", "ppg.h"),
]
_SOURCES = _KERNEL_SOURCES + [os.path.join(EMU_DIR, "ppg_emu.cpp"), os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_host.h"),
                              os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_pack.h"),
                              os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_fetch.h")]
_lib = None


def _stale(out, sources):
    return not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in sources)


def asan_runtime():
    """Path of gcc's libasan.so: an AddressSanitizer build can only be loaded into a process that has it preloaded (LD_PRELOAD)."""
    return subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()


def build(sanitize=False):
    """libppg_emu.so (libppg_emu_ubsan.so / libppg_emu_asan.so): the dispatch / C-ABI unit and twelve kernel units (family x prey
    registers), compiled in parallel (one translation unit took six minutes under UBSan).  sanitize: False, True (= "undefined") or
    "address" -- the AddressSanitizer build also poisons the bytes behind a workgroup's LDS (wave_emu.h), so that an out-of-bounds
    LDS READ traps too (round 3's advisor found one that UBSan cannot see)."""
    asan = sanitize == "address"
    out = EMU_LIB if not sanitize else EMU_LIB.replace(".so", "_asan.so" if asan else "_ubsan.so")
    if not _stale(out, _SOURCES):
        return out
    from concurrent.futures import ThreadPoolExecutor
    tag = f"{'asan' if asan else 'ubsan' if sanitize else 'emu'}.{os.getpid()}"   # concurrent builders (multi-process tests) never share a file
    objdir = os.path.join(EMU_DIR, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-std=c++17", "-ffp-contract=off", "-fno-omit-frame-pointer", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-fPIC"]
    san_link = ["-fsanitize=address"] if asan else ["-fsanitize=undefined"] if sanitize else []
    flags += ["-O1", "-g", "-fsanitize=address"] if asan else ["-O1", "-fsanitize=undefined", "-fno-sanitize-recover=undefined"] if sanitize else ["-O2", "-g"]
    units = [("ppg_emu.cpp", [], "main")] + [("ppg_emu_part.cpp", [f"-DPPG_EMU_FAMILY={f}", f"-DPPG_EMU_NQ={q}"], f"f{f}_q{q}")
                                            for f in (0, 1, 2, 3) for q in (4, 2, 1)]
    cmds = [(["g++", *flags, *defs, "-c", "-o", os.path.join(objdir, f"{name}.{tag}.o"), os.path.join(EMU_DIR, src)])
            for src, defs, name in units]
    objs = [c[-2] for c in cmds]
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) - 1))) as pool:
            for r in pool.map(lambda c: subprocess.run(c), cmds):
                r.check_returncode()
        tmp = f"{out}.{os.getpid()}.tmp"
        subprocess.run(["g++", "-shared", "-o", tmp, *objs] + san_link, check=True)
        os.replace(tmp, out)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    return out


def library(sanitize=False):
    global _lib
    from predpreygrass_amd import _abi
    if sanitize:  # UBSan build: traps (aborts the process) on signed overflow, bad shifts, misaligned access ...; "address": ASan (the
        # process must have been started with LD_PRELOAD=asan_runtime())
        return _abi.bind(ctypes.CDLL(build(sanitize=sanitize)))
    if _lib is None:
        _lib = _abi.bind(ctypes.CDLL(build()))
    return _lib
