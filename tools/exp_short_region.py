"""GPU experiment: the driver's timed region (sync, 20 steps on 3 sub-batch streams, sync) repeated many times in ONE process:
wall time per step with (a) no events, (b) torch timing events (system-scope release), (c) raw HIP events created with
hipEventReleaseToDevice, (d) events + polling for completion.   usage: PYTHONPATH=. python tools/exp_short_region.py [reps=40]"""
import ctypes, sys, time, statistics
import torch
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventQuery.argtypes = [ctypes.c_void_p]


class RawEvent:
    def __init__(self, flags):
        self.h = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(self.h), flags) == 0

    def record(self, stream):
        assert hip.hipEventRecord(self.h, ctypes.c_void_p(stream.cuda_stream)) == 0

    def elapsed_time(self, other):
        ms = ctypes.c_float()
        assert hip.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h) == 0
        return ms.value

    def query(self):
        return hip.hipEventQuery(self.h) == 0


g = SubBatchedPredPreyGrass({}, batch_size=4096, n_sub=3, device=dev, obs_dtype=torch.float64, seed=0)
g.reset()
for i in range(3072):
    if i % 64 == 0:
        g.rebalance()
    g.step(random_actions=True, auto_reset=True)
torch.cuda.synchronize()
MODES = {"none": None, "torch": lambda: torch.cuda.Event(enable_timing=True), "raw_default": lambda: RawEvent(0),
         "raw_release_to_device": lambda: RawEvent(0x40000000), "torch+poll": lambda: torch.cuda.Event(enable_timing=True)}
res = {m: ([], []) for m in MODES}
for r in range(reps):
    for mode, mk in MODES.items():
        for _ in range(5):
            g.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        e0 = [mk() for _ in g.streams] if mk else []
        e1 = [mk() for _ in g.streams] if mk else []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, e in zip(g.streams, e0):
            e.record(s)
        for _ in range(20):
            g.step(random_actions=True, auto_reset=True)
        for s, e in zip(g.streams, e1):
            e.record(s)
        if mode == "torch+poll":
            while not all(e.query() for e in e1):
                pass
        torch.cuda.synchronize()
        res[mode][0].append((time.perf_counter() - t0) * 1e6 / 20)
        if mk:
            res[mode][1].append(sum(a.elapsed_time(b) for a, b in zip(e0, e1)) / len(e0) * 1e3 / 20)
for mode, (w, e) in res.items():
    print(f"{mode:24s} wall/step us median {statistics.median(w):6.1f} min {min(w):6.1f} | events/step median "
          f"{statistics.median(e) if e else 0:6.1f}")
