#!/usr/bin/env python3
"""usage: gpu_sweep_repro.py seed [seed ...]: the walls leg of tools/gpu_sweep.py for single seeds, with the configuration printed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from predpreygrass_amd.walls_occlusion import PredPreyGrass as WOEnv
from tests import test_rq_random_configs as T2
for seed in [int(a) for a in sys.argv[1:]]:
    made = []
    def mk(cfg):
        e = WOEnv(cfg, device="cuda:0")
        made.append((cfg, e))
        return e
    try:
        T2.run_differential(mk, seed, walls=True)
        print(seed, "ok")
    except AssertionError as ex:
        cfg, e = made[-1]
        print(seed, "FAIL", repr(ex)[:120], {k: cfg[k] for k in cfg if "obs_range" in k or k in ("grid_size", "include_visibility_channel", "mask_observation_with_visibility", "respect_los_for_movement", "num_walls")},
              "kernel", e._env.step_kernel_name() if hasattr(e, "_env") else None, flush=True)
