"""GPU experiment (not product code): does the step time depend on WHICH buffers the envs live in?  Several independent env groups
(own tensors each) in one process, timed round-robin in short segments: the same kernel, the same clocks, the same minute.

    python tools/exp_placement.py [--groups 4] [--rounds 4] [--segment 300] [--streams 3]
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from predpreygrass_amd.config import config_env  # noqa: E402
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, default=4)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--segment", type=int, default=300)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--preroll", type=int, default=1024)
ap.add_argument("--spacer-mb", type=int, default=0, help="allocate (and keep) this many MB between two groups")
ap.add_argument("--modes", default="torch", type=lambda v: v.split(","),
                help="how group g allocates its observation tensors, cycled: torch (caching allocator), hipmalloc (hipExtMallocWithFlags default), "
                     "contiguous (hipDeviceMallocContiguous: physically contiguous)")
ap.add_argument("--prey-caps", default="", type=lambda v: [int(x) for x in v.split(",")] if v else [],
                help="prey row capacity of group g, cycled (default: the library's 128): the slab stride of obs_prey is capacity x 2592 bytes")
ap.add_argument("--swap", action="store_true", help="second pass: NEW groups (fresh row tables, env words, handles) on the OLD groups' observation tensors")
ap.add_argument("--arena-gb", type=float, default=0, help="first allocate one block of this size and give it back to torch's caching allocator: "
                                                        "every later tensor is then carved out of that ONE device allocation")
args = ap.parse_args()
if args.arena_gb:
    arena = torch.empty(int(args.arena_gb * (1 << 30)), dtype=torch.uint8, device="cuda:0")
    print("arena at", hex(arena.data_ptr()))
    del arena

import ctypes  # noqa: E402

from predpreygrass_amd.batched import BatchedPredPreyGrass  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]


class RawDeviceBuffer:
    """hipExtMallocWithFlags memory as a torch tensor (never freed: an experiment)."""
    def __init__(self, nbytes, flags):
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, flags)
        if rc != 0 or not p.value:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flags}) = {rc}")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}


class MemLocation(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class AllocFlags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class MemAllocationProp(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", MemLocation),
                ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", AllocFlags)]


class MemAccessDesc(ctypes.Structure):
    _fields_ = [("location", MemLocation), ("flags", ctypes.c_int)]


hip.hipMemAddressReserve.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
hip.hipMemCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.POINTER(MemAllocationProp), ctypes.c_ulonglong]
hip.hipMemMap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
hip.hipMemSetAccess.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(MemAccessDesc), ctypes.c_size_t]
hip.hipMemGetAllocationGranularity.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(MemAllocationProp), ctypes.c_int]


hip.hipMemRelease.argtypes = [ctypes.c_void_p]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]


class ScrambledDeviceBuffer:
    """A VA-contiguous buffer whose physical chunks (`chunk` bytes each, own hipMemCreate) are mapped in a pseudo-random order
    (or in allocation order: scramble=False).  spread = N > 1: N times as many chunks are created, a random n of them are mapped
    and the others released again -- the buffer's physical pages then come from an N times larger stretch of device memory."""
    def __init__(self, nbytes, chunk=2 << 20, scramble=True, seed=1, spread=1, regions=1, gap_gb=0):
        prop = MemAllocationProp()
        prop.type = 1                    # hipMemAllocationTypePinned
        prop.location = MemLocation(1, 0)   # device 0
        gran = ctypes.c_size_t()
        rc = hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 1)
        assert rc == 0, rc
        chunk = max(chunk, gran.value) // gran.value * gran.value
        n = (nbytes + chunk - 1) // chunk
        size = n * chunk
        base = ctypes.c_void_p()
        rc = hip.hipMemAddressReserve(ctypes.byref(base), size, 2 << 20, None, 0)
        assert rc == 0 and base.value, rc
        import random
        rng = random.Random(seed)
        order = list(range(n))
        if scramble:
            rng.shuffle(order)
        handles, spacers = [], []
        per_region = (n * spread + regions - 1) // regions
        for i in range(n * spread):
            if regions > 1 and i and i % per_region == 0:   # a spacer allocation between two parts of the pool: the next chunks come from beyond it
                sp = ctypes.c_void_p()
                if hip.hipMalloc(ctypes.byref(sp), int(gap_gb * (1 << 30))) == 0:
                    spacers.append(sp)
            h = ctypes.c_void_p()
            rc = hip.hipMemCreate(ctypes.byref(h), chunk, ctypes.byref(prop), 0)
            assert rc == 0, ("hipMemCreate", rc, i)
            handles.append(h)
        for sp in spacers:
            hip.hipFree(sp)
        pick = sorted(rng.sample(range(n * spread), n)) if spread > 1 else list(range(n))
        chosen = set(pick)
        for j, i in enumerate(pick):   # the j-th chosen chunk (allocation order) goes to slot order[j] of the virtual range
            rc = hip.hipMemMap(ctypes.c_void_p(base.value + order[j] * chunk), chunk, 0, handles[i], 0)
            assert rc == 0, ("hipMemMap", rc, i)
        for i, h in enumerate(handles):
            if i not in chosen:
                hip.hipMemRelease(h)
        acc = MemAccessDesc(MemLocation(1, 0), 3)
        rc = hip.hipMemSetAccess(base, size, ctypes.byref(acc), 1)
        assert rc == 0, ("hipMemSetAccess", rc)
        self.ptr, self.granularity, self.chunks = base.value, gran.value, n
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}


_orig_alloc = BatchedPredPreyGrass._alloc_buffers
alloc_mode = ["torch"]


def _alloc_with_mode(self, prey_capacity):
    _orig_alloc(self, prey_capacity)
    if alloc_mode[0] == "torch":
        return
    for name in ("obs_pred", "obs_prey"):
        t = getattr(self, name)
        nbytes = t.numel() * t.element_size()
        m = alloc_mode[0]
        if m.startswith("scrambled") or m.startswith("ordered"):   # scrambled[:chunk KB] / ordered[:chunk KB]
            kb = int(m.split(":")[1]) if ":" in m else 2048          # chunk size in KB
            f = m.split(":")                                                              # scrambled:<chunk KB>:<spread>[:<regions>:<gap GB>]
            spread = int(f[2]) if len(f) > 2 else 1
            regions, gap = (int(f[3]), float(f[4])) if len(f) > 4 else (1, 0)
            raw = ScrambledDeviceBuffer(nbytes, chunk=kb << 10, scramble=m.startswith("scrambled"), seed=id(self) & 0xFFFF, spread=spread,
                                        regions=regions, gap_gb=gap)
            if not hasattr(_alloc_with_mode, "said"):
                _alloc_with_mode.said = print(f"({m}: {raw.chunks} chunks, granularity {raw.granularity})")
        elif m.endswith("_off"):   # contiguous_off / hipmalloc_off: the tensor starts at a per-tensor pseudo-random multiple of 4 KB inside its allocation
            _alloc_with_mode.count = getattr(_alloc_with_mode, "count", 0) + 1
            delta = ((_alloc_with_mode.count * 2654435761) >> 7) % 1024 * 4096          # < 4 MB
            raw = RawDeviceBuffer(nbytes + (4 << 20), {"contiguous_off": 0x4, "hipmalloc_off": 0x0}[m])
            raw.ptr += delta
            raw.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (raw.ptr, False), "version": 2}
        else:
            raw = RawDeviceBuffer(nbytes, {"contiguous": 0x4, "hipmalloc": 0x0}[m])
        new = torch.as_tensor(raw, device="cuda:0").view(t.dtype).view(t.shape)
        new.zero_()
        setattr(self, name, new)
        setattr(self, "_raw_" + name, raw)


BatchedPredPreyGrass._alloc_buffers = _alloc_with_mode
ap2_modes = None
print("gpu", torch.cuda.get_device_properties(0).uuid)
groups, spacers = [], []
for g in range(args.groups):
    alloc_mode[0] = args.modes[g % len(args.modes)]
    kw = {"prey_capacity": args.prey_caps[g % len(args.prey_caps)]} if args.prey_caps else {}
    t_alloc = time.perf_counter()
    grp = SubBatchedPredPreyGrass(dict(config_env), batch_size=args.envs, n_sub=args.streams, device="cuda:0", obs_dtype=torch.float64, seed=1000 * g, **kw)
    grp.reset()
    torch.cuda.synchronize()
    print(f"group {g} [{alloc_mode[0]}] built in {time.perf_counter() - t_alloc:.2f} s")
    groups.append(grp)
    if args.spacer_mb:
        spacers.append(torch.empty(args.spacer_mb << 20, dtype=torch.uint8, device="cuda:0"))
step_no = [0] * args.groups


def run(g, n):
    grp = groups[g]
    for _ in range(n):
        if step_no[g] % 64 == 0:
            grp.rebalance()
        step_no[g] += 1
        grp.step(random_actions=True, auto_reset=True)


for g in range(args.groups):
    run(g, args.preroll)
torch.cuda.synchronize()
res = [[] for _ in groups]
for r in range(args.rounds):
    for g in (range(args.groups) if r % 2 == 0 else reversed(range(args.groups))):
        run(g, 20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(g, args.segment)
        torch.cuda.synchronize()
        res[g].append((time.perf_counter() - t0) / args.segment * 1e6)
print(f"# {args.groups} env groups of {args.envs} envs ({args.streams} sub-batches each), {args.segment}-step segments, us per step")
for g, v in enumerate(res):
    ptrs = [hex(e.obs_prey.data_ptr()) for e in groups[g].subs]
    print(f"group {g} [{args.modes[g % len(args.modes)]:10s} cap {groups[g].subs[0].prey_capacity:3d}]: median {statistics.median(v):7.2f}  all {[round(x, 1) for x in v]}   obs_prey at {ptrs}")

# does a plain fill of the same buffers see the same difference?  (a probe a constructor could run)
def fill_time(tensors, reps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for t in tensors:
            t.zero_()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(t.numel() * t.element_size() for t in tensors) * reps / dt / 1e12


for g, grp in enumerate(groups):
    big = [t for e in grp.subs for t in (e.obs_pred, e.obs_prey)]
    bw = [fill_time(big) for _ in range(3)]
    each = [round(fill_time([t], 60), 2) for t in big]
    print(f"group {g}: fill of its six observation tensors {[round(b, 2) for b in bw]} TB/s;  one at a time {each}")

if args.swap:
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    donors = [[(e.obs_pred, e.obs_prey) for e in grp.subs] for grp in groups]
    old = [statistics.median(v) for v in res]
    for grp in groups:
        for e in grp.subs:
            e.close()
    orig = BatchedPredPreyGrass._alloc_buffers
    pool = []

    def patched(self, prey_capacity):
        orig(self, prey_capacity)
        self.obs_pred, self.obs_prey = pool.pop(0)   # (the freshly allocated pair is dropped: the donor's tensors are used)

    BatchedPredPreyGrass._alloc_buffers = patched
    new_groups = []
    for g in reversed(range(args.groups)):   # (created in the opposite order: their tables land somewhere else)
        pool.extend(donors[g])
        grp = SubBatchedPredPreyGrass(dict(config_env), batch_size=args.envs, n_sub=args.streams, device="cuda:0", obs_dtype=torch.float64, seed=77 + g)
        grp.reset()
        new_groups.append((g, grp))
    new_groups.sort()
    groups[:] = [grp for _, grp in new_groups]
    step_no[:] = [0] * args.groups
    for g in range(args.groups):
        run(g, args.preroll)
    torch.cuda.synchronize()
    res2 = [[] for _ in groups]
    for r in range(args.rounds):
        for g in (range(args.groups) if r % 2 == 0 else reversed(range(args.groups))):
            run(g, 20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(g, args.segment)
            torch.cuda.synchronize()
            res2[g].append((time.perf_counter() - t0) / args.segment * 1e6)
    print("# second pass: new handles, row tables and env words; group g writes into the observation tensors group g had in the first pass")
    for g, v in enumerate(res2):
        print(f"group {g}: first pass {old[g]:7.2f}   new tables on the same observation tensors {statistics.median(v):7.2f}  {[round(x, 1) for x in v]}")

# the same, for the part of every env's slab the step writes (the first rows of each env): strided fills
for g, grp in enumerate(groups):
    part = [t for e in grp.subs for t in (e.obs_pred[:, :6], e.obs_prey[:, :24])]
    bw = [fill_time(part, 60) for _ in range(3)]
    print(f"group {g}: fill of the first 6 / 24 rows of every env's slab {[round(b, 2) for b in bw]} TB/s   (step: {statistics.median(res[g]):.1f} us)")
