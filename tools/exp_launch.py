#!/usr/bin/env python3
"""Where does a short step go: host launch cost vs kernel duration (uses an ablation build if given)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from predpreygrass_amd import _abi
if len(sys.argv) > 1:
    _abi._lib = _abi.bind(ctypes.CDLL(sys.argv[1]))
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
for B in (64, 4096):
    env = BatchedPredPreyGrass(config_env, batch_size=B, device="cuda:0")
    env.reset()
    for _ in range(300): env.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    # (1) host issue time: 200 launches, then wait
    t0 = time.perf_counter()
    for _ in range(200): env.step(random_actions=True, auto_reset=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    # (2) single-kernel latency: launch + sync each
    lat = []
    for _ in range(200):
        a = time.perf_counter(); env.step(random_actions=True, auto_reset=True); torch.cuda.synchronize(); lat.append(time.perf_counter() - a)
    # (3) kernel duration by events around each launch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a, b in evs:
        a.record(); env.step(random_actions=True, auto_reset=True); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"B={B}: host issue {1e6*(t1-t0)/200:.1f} us/launch, issue+drain {1e6*(t2-t0)/200:.1f} us/step, "
          f"launch+sync latency median {1e6*sorted(lat)[100]:.1f} us, event-bracketed kernel median {d[100]:.1f} us", flush=True)
