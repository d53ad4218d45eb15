#!/usr/bin/env python3
"""Long randomised differential sweep on the GPU: random configurations of the base family, the second generation and
the walls variant, dict API vs the CPU oracles, call by call, bit for bit (round 6: the second-generation classes run with their analytics
mirror cross-checking every energy it derives against the device's).  usage: gpu_sweep.py <first seed> <n seeds>"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from predpreygrass_amd.env import PredPreyGrass
from predpreygrass_amd.red_queen import PredPreyGrass as RQEnv
from predpreygrass_amd.walls_occlusion import PredPreyGrass as WOEnv
from tests import test_random_configs as T1
from tests import test_rq_random_configs as T2

first, n = int(sys.argv[1]), int(sys.argv[2])
fails, t0 = [], time.time()
counts = {"base": 0, "gen2": 0, "walls": 0}
for seed in range(first, first + n):
    for kind, fn in (("base", lambda: T1.run_differential(lambda cfg: PredPreyGrass(cfg, device="cuda:0"), seed)),
                     ("gen2", lambda: T2.run_differential(lambda cfg: RQEnv(cfg, device="cuda:0", _check_analytics=True), seed)),
                     ("walls", lambda: T2.run_differential(lambda cfg: WOEnv(cfg, device="cuda:0", _check_analytics=True), seed, walls=True))):
        try:
            fn()
            counts[kind] += 1
        except Exception as ex:  # noqa: BLE001
            fails.append((kind, seed, repr(ex)[:400]))
            traceback.print_exc()
            if len(fails) > 10:
                break
    if (seed - first) % 20 == 19:   # (progress: a run that is cut off still says what it covered)
        print("progress", counts, "fails", len(fails), f"{time.time() - t0:.0f} s", flush=True)
print("ok", counts, "fails", fails, f"{time.time() - t0:.0f} s")
