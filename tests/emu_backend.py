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
_SOURCES = [
    os.path.join(EMU_DIR, "ppg_emu.cpp"), os.path.join(EMU_DIR, "wave_emu.h"),
    os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_kernel.h"),
    os.path.join(ROOT, "predpreygrass_amd", "csrc", "ppg_host.h"),
    os.path.join(ROOT, "include", "ppg.h"),
]
_lib = None


def build(sanitize=False):
    out = EMU_LIB if not sanitize else EMU_LIB.replace(".so", "_ubsan.so")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(s) for s in _SOURCES):
        return out
    tmp = f"{out}.{os.getpid()}.tmp"   # concurrent builders (multi-process tests) must never see a half-written library
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fno-omit-frame-pointer", "-Wall",
           "-Wno-unknown-pragmas", "-Wno-unused-function", "-fPIC", "-shared", "-o", tmp, _SOURCES[0]]
    if sanitize:
        cmd[3:3] = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


def library(sanitize=False):
    global _lib
    from predpreygrass_amd import _abi
    if sanitize:  # UBSan build: traps (aborts the process) on signed overflow, bad shifts, misaligned access ...
        return _abi.bind(ctypes.CDLL(build(sanitize=True)))
    if _lib is None:
        _lib = _abi.bind(ctypes.CDLL(build()))
    return _lib
