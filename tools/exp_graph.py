#!/usr/bin/env python3
"""Launch-bound regime: eager ppg_step calls vs a captured hipGraph of 16 steps, small batches (us per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env

for B in (64, 256, 1024):
    env = BatchedPredPreyGrass(dict(config_env), batch_size=B, device="cuda:0")
    env.reset(seed=1)
    for _ in range(200):
        env.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(1600):
        env.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / 1600 * 1e6
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for _ in range(16):
            env.step(random_actions=True, auto_reset=True)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / 1600 * 1e6
    print(f"{B} envs: eager {eager:.1f} us/step ({B / eager:.2f} M env-steps/s), hipGraph of 16 steps {graph:.1f} us/step "
          f"({B / graph:.2f} M env-steps/s)", flush=True)
