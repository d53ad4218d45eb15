#!/usr/bin/env python3
"""Throughput of ppg_rollout (K fused steps per launch) vs K single-step launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
from predpreygrass_amd.config import config_env
B = 4096
for n_sub in (1, 2, 3):
    for K in (1, 8, 32, 128):
        g = SubBatchedPredPreyGrass(config_env, batch_size=B, n_sub=n_sub, device="cuda:0", seed=0)
        g.reset(); g.synchronize()
        g.rollout(320, random_actions=True, auto_reset=True); torch.cuda.synchronize()
        n_launch = max(1, 2048 // K)
        t0 = time.perf_counter()
        for _ in range(n_launch): g.rollout(K, random_actions=True, auto_reset=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = n_launch * K
        print(f"sub-batches={n_sub} K={K:4d}: {dt/steps*1e6:7.1f} us per 4096-env step  {B*steps/dt/1e6:6.1f} M env-steps/s", flush=True)
        del g
