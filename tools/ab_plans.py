"""GPU experiment helper (not product code): interleaved A/B of step-kernel wave plans IN ONE PROCESS on the SAME envs.

The wave plan (ppg_set_wave_plan) only decides which kernel steps the envs, never a result, so the plans can be switched between
short timed segments of one long rollout: every plan sees the same population, the same box and the same clock drift.

    python tools/ab_plans.py [--streams 3] [--rounds 6] [--segment 300] [--workload base|c4] [--envs 4096] PLAN [PLAN ...]
    PLAN = name:waves,helper_min_rows,coop_envs        e.g.  pair:2,0,0  coop44:4,0,4  auto:0,0,0
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
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--segment", type=int, default=300)
ap.add_argument("--preroll", type=int, default=3072)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--workload", default="base")
ap.add_argument("--obs-dtype", default="f64")
ap.add_argument("--rebalance-every", type=int, default=64)
ap.add_argument("--obs-spread", type=int, default=32, help="BatchedPredPreyGrass(obs_spread=N): observation tensors on spread physical pages (0 = torch's allocator)")
ap.add_argument("plans", nargs="+")
args = ap.parse_args()

cfg = dict(config_env)
if args.workload == "c4":
    cfg.update({"grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32, "predator_obs_range": 7, "prey_obs_range": 7})
plans = []
for spec in args.plans:
    name, v = spec.split(":")
    plans.append((name, [int(x) for x in v.split(",")]))
kw = {}
if args.workload == "red_queen":   # the second-generation env on its reference config, float32 observations like the reference
    from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
    cfg, kw = dict(config_env_base), {"env_class": BatchedRedQueen}
    if "--obs-dtype" not in sys.argv:
        args.obs_dtype = "f32"
if args.workload == "walls":       # the walls variant: the reference's zigzag layout with every line-of-sight option on
    from predpreygrass_amd.red_queen import BatchedRedQueen
    from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls
    cfg, kw = dict(config_env_zigzag_walls), {"env_class": BatchedRedQueen, "walls": True}
    if "--obs-dtype" not in sys.argv:
        args.obs_dtype = "f32"
if args.workload == "drive":
    cfg["enable_drive_channels"] = True
group = SubBatchedPredPreyGrass(cfg, batch_size=args.envs, n_sub=args.streams, device="cuda:0",
                                obs_dtype=torch.float64 if args.obs_dtype == "f64" else torch.float32, obs_spread=args.obs_spread, **kw)
if args.workload == "walls":
    for e in group.subs:
        e.set_walls(cfg["manual_wall_positions"])
group.reset()
step_no = 0


def run(n):
    global step_no
    for _ in range(n):
        if args.rebalance_every > 0 and step_no % args.rebalance_every == 0:
            group.rebalance()
        step_no += 1
        group.step(random_actions=True, auto_reset=True)


run(args.preroll)
torch.cuda.synchronize()
res = {name: [] for name, _ in plans}
kernels = {}
for r in range(args.rounds):
    order = plans if r % 2 == 0 else plans[::-1]
    for name, wp in order:
        for e in group.subs:
            e.set_wave_plan(*wp)
        kernels[name] = group.subs[0].step_kernel_name() + " " + str(group.subs[0].wave_plan())
        run(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.segment)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / args.segment * 1e6)
print(f"# {args.workload} {args.envs} envs, {args.streams} streams, segments of {args.segment} steps, us per full step (all sub-batches)")
for name, v in res.items():
    med = statistics.median(v)
    print(f"{name:14s} median {med:7.2f} us = {args.envs / med:6.2f} M env-steps/s   min {min(v):7.2f}   all {[round(x, 1) for x in v]}   {kernels[name]}")
