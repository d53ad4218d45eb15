#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched PredPreyGrass step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE batched call of the hot path: every env of the shard advances by one
transition (predpreygrass_rllib_env.py:219-473 of the reference) -- device-side uniform random
actions (Philox4x32-10), decay, grass regrowth, ordered movement, engagement, reproduction and
float64 observations written for every agent -- with auto-reset of finished episodes.  State
and observation buffers are resident in HBM; nothing crosses PCIe inside the timed region.

Workload (BASELINE.json configs[2], the headline): 4096 envs x 25x25 grid per GPU, default
config (6 predators / 8 prey / 100 grass, obs 7x7 / 9x9).  N GPUs = N independent shards of 4096
envs (weak scaling); with --gather (default for N>1) each step is followed by the RCCL all-gather
of the compacted observation tensors that north_star specifies.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def cpu_baseline(cfg, seed0, seconds=12.0, threads=None):
    """The CPU oracle (C restatement of the reference step(), kind "port") timed on this host on a
    bounded sample of the same workload: one env per thread, random actions, auto-reset."""
    import concurrent.futures as cf
    from oracle.ppg_oracle import OracleEnv
    threads = threads or os.cpu_count() or 1
    chunk = 2000

    def work(i):
        env = OracleEnv(cfg)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            n += env.rollout_random(seed0 + i, chunk)
        return n, time.perf_counter() - t0

    # single thread first (the scalar port), then all cores (ctypes releases the GIL)
    n1, t1 = work(0)
    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(work, range(threads)))
    tot = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {
        "value": round(tot / wall, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
        "single_thread_value": round(n1 / t1, 1),
        "sample": f"{threads} threads x 1 env each, default config, Philox random actions + auto-reset, "
                  f"{seconds:.0f} s per thread ({tot} env-steps); oracle/ppg_oracle.c (-O2 -ffp-contract=off)",
        "reference_python_fixed": "reference Python step(): 45 env-steps/s on 1 core, ~400 on 8 cores "
                                  "(measured in the survey container, BASELINE.md section 2)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--obs-dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--gather", dest="gather", action="store_true", default=None)
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank if distributed else 0)
    n_gpus = world if distributed else 1
    gather = args.gather if args.gather is not None else distributed

    cfg = dict(config_env)
    B = args.envs
    obs_dtype = torch.float64 if args.obs_dtype == "f64" else torch.float32
    env = BatchedPredPreyGrass(cfg, batch_size=B, device=device, obs_dtype=obs_dtype,
                               seed=args.seed + rank * B)
    env.reset()
    gatherer = None
    if gather and distributed:
        from predpreygrass_amd.distributed import ObservationGatherer
        gatherer = ObservationGatherer(env)

    def one_step():
        env.step(random_actions=True, auto_reset=True)
        if gatherer is not None:
            gatherer.gather()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(device)
    # zero the observation counters (bandwidth accounting) -- outside the timed region
    env.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1] = 0
    calls0 = env.env_state[:, _abi.ENV_CALLS].clone()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step()
    ev1.record()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if distributed:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- accounting ------------------------------------------------------------------
    es = env.env_state.cpu().numpy().astype("int64")
    assert ((es[:, _abi.ENV_CALLS] - calls0.cpu().numpy()) == args.steps).all()
    status = int((es[:, _abi.ENV_STATUS]).max())
    n_obs_pred = int(es[:, _abi.ENV_OBS_PRED].sum())
    n_obs_prey = int(es[:, _abi.ENV_OBS_PREY].sum())
    G, Rp, Rq = env.grid_size, env.Rp, env.Rq
    osz = 8 if obs_dtype == torch.float64 else 4
    env_steps_rank = B * args.steps
    # 2*(3*G^2*8) + sum_obs(4*R^2*osz) + 64*L per env-step, L = rows in use
    alg_bytes = env_steps_rank * 2 * 3 * G * G * 8 + n_obs_pred * 4 * Rp * Rp * osz + \
        n_obs_prey * 4 * Rq * Rq * osz + 64 * (n_obs_pred + n_obs_prey)
    # bytes this implementation has to move at minimum (no dense grid exists in HBM): observations +
    # row tables r/w (27 B read, 35 B written per row) + grass table (8 B r/w, 2 B read) + env words
    min_bytes = n_obs_pred * 4 * Rp * Rp * osz + n_obs_prey * 4 * Rq * Rq * osz + \
        62 * (n_obs_pred + n_obs_prey) + env_steps_rank * (env.n_grass * 18 + 2 * 64 + 8)
    kernel_s = dev_ms / 1e3 / args.steps
    achieved = alg_bytes / args.steps / kernel_s / 1e9
    value = n_gpus * env_steps_rank / wall

    if rank == 0:
        out = {
            "metric": "env-steps/sec at 4096x(25x25) grids, 1/2/4/8 MI355X; % HBM roofline",
            "value": round(value, 1),
            "unit": "env-steps/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{B} envs x {G}x{G} grid per GPU, default config (6 predators / 8 prey / 100 grass, "
                            f"obs {Rp}x{Rp} / {Rq}x{Rq} {args.obs_dtype}), device-side uniform random actions, "
                            "auto-reset, observations written every step (BASELINE.json configs[2])",
                "envs_per_gpu": B,
                "parallelism": f"batch-sharded x{n_gpus}" + (", RCCL all-gather of observations" if gatherer else ""),
                "mean_agents_per_env": round((n_obs_pred + n_obs_prey) / env_steps_rank, 2),
                "status_bits": status,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "kernel": "ppg_step_q2",
                "kernel_ms": round(kernel_s * 1e3, 5),
                "algorithmic_bytes_per_launch": int(alg_bytes / args.steps),
                "implementation_min_bytes_per_launch": int(min_bytes / args.steps),
                "note": "achieved = SURVEY 8(d) algorithmic bytes / mean launch-to-launch time (HIP events on "
                        "the launch stream). The dense grid term of that formula is never moved by this design; "
                        "implementation_min_bytes_per_launch is what the kernel must actually touch.",
            },
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, seconds=args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
