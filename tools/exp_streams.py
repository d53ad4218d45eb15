#!/usr/bin/env python3
"""Experiment: B envs as n independent sub-batches on n HIP streams (phases de-synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
B = 4096
for n in [1, 2, 4, 8]:
    envs = [BatchedPredPreyGrass(config_env, batch_size=B // n, device="cuda:0", seed=i * (B // n)) for i in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    for e in envs: e.reset()
    torch.cuda.synchronize()
    def run(k):
        for _ in range(k):
            for e, s in zip(envs, streams):
                with torch.cuda.stream(s):
                    e.step(random_actions=True, auto_reset=True)
    run(300); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 2000; run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n_streams={n}: {B*K/dt/1e6:.2f} M env-steps/s, {dt/K*1e6:.1f} us per full step", flush=True)
    del envs
