"""GPU experiment: the fused cooperative rollout (ppg_rollout on a cooperative plan: ppgc_rollout) against per-step launches.
    python tools/exp_coop_rollout.py [envs=4096] [streams=3] [n=100]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predpreygrass_amd.config import config_env
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for streams in (1, S):
    grp = SubBatchedPredPreyGrass(dict(config_env), batch_size=B, n_sub=streams, device="cuda:0")
    grp.reset()
    for _ in range(3000):
        grp.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    res = {"steps": [], "fused": []}
    for r in range(5):
        t0 = time.perf_counter()
        for _ in range(3 * N):
            grp.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        res["steps"].append((time.perf_counter() - t0) / (3 * N) * 1e6)
        t0 = time.perf_counter()
        for _ in range(3):
            grp.rollout(N, random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        res["fused"].append((time.perf_counter() - t0) / (3 * N) * 1e6)
    for k, v in res.items():
        v.sort()
        print(f"{B} envs, {streams} sub-batches, {k:6s}: median {v[len(v)//2]:6.2f} us per step = {B / v[len(v)//2]:6.2f} M env-steps/s   all {[round(x, 1) for x in v]}")
