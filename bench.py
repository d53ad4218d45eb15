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
envs (weak scaling, no data-path collective: envs never interact).  For N>1 an extra leg measures
the step followed by the RCCL all-gather of the compacted observations that north_star specifies
and reports it under "obs_gather" (it is bandwidth-bound on xGMI).  Within one GPU the
4096 envs are stepped as --streams (default 3) independent sub-batches on separate HIP streams.

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


def cpu_baseline(cfg, seed0, seconds=12.0, threads=None, workload="base"):
    """The CPU oracle (C restatement of the reference step(), kind "port") timed on this host on a
    bounded sample of the same workload: one env per thread, random actions, auto-reset."""
    import concurrent.futures as cf
    if workload == "red_queen":
        from oracle.rq_oracle import RQOracleEnv as OracleEnv
    else:
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
        "sample": f"{threads} threads x 1 env each, {'red_queen config_env_base' if workload == 'red_queen' else 'default config'}, "
                  f"Philox random actions + auto-reset, {seconds:.0f} s per thread ({tot} env-steps); "
                  f"oracle/{'rq_oracle.c' if workload == 'red_queen' else 'ppg_oracle.c'} (-O2 -ffp-contract=off)",
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
    ap.add_argument("--obs-dtype", choices=["f64", "f32"], default=None,
                    help="observation dtype (default: f64 for the base workload, f32 -- the reference's -- for red_queen)")
    ap.add_argument("--workload", choices=["base", "c4", "red_queen", "drive", "walls"], default="base",
                    help="base: BASELINE.json configs[2] (the headline); c4: configs[3] (64x64 grid, 16 predators / 32 prey, "
                         "7x7 windows); red_queen: the second-generation env (SURVEY 8(f) N2) with its reference config; "
                         "drive: the drive-conditioned variant of the default config; walls: the walls variant with the "
                         "reference's zigzag layout and every line-of-sight option on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=3,
                    help="independent sub-batches per GPU, each on its own HIP stream (1 = one launch per step)")
    ap.add_argument("--rebalance-every", type=int, default=64,
                    help="call ppg_rebalance every that many steps (0 = never): heavy envs are assigned to workgroups first")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank, so that the gather legs run (needs torchrun)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY: run the control flow on CPU (gloo, wave-emulator build of the kernel); numbers are meaningless")
    ap.add_argument("--gather-steps", type=int, default=50,
                    help="N>1 only: extra leg of this many steps with the RCCL observation all-gather after each step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist   # --force-dist: exercise the RCCL legs with a world of one rank (test hook)
    dry = args.dry_run_cpu
    if dry:
        from tests.emu_backend import library as _emu_library
        if distributed:
            dist.init_process_group("gloo")
        device = torch.device("cpu")

        class _NoEvent:
            def __init__(self, **kw):
                self.t = 0.0

            def record(self, stream=None):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3

        torch.cuda.synchronize = lambda *a, **k: None   # nothing asynchronous on the CPU path
        torch.cuda.Event = _NoEvent
    else:
        if distributed:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = torch.device("cuda", local_rank if distributed else 0)
        torch.cuda.set_device(device)
    n_gpus = world if distributed else 1

    rq = args.workload in ("red_queen", "walls")
    extra_kw = {}
    if args.obs_dtype is None:
        args.obs_dtype = "f32" if rq else "f64"
    if args.workload == "walls":
        from predpreygrass_amd.red_queen import BatchedRedQueen
        from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls
        cfg, env_class, extra_kw = dict(config_env_zigzag_walls), BatchedRedQueen, {"walls": True}
    elif rq:
        from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
        cfg, env_class = dict(config_env_base), BatchedRedQueen
    else:
        cfg, env_class = dict(config_env), BatchedPredPreyGrass
        if args.workload == "c4":   # BASELINE.json configs[3]
            cfg.update({"grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32,
                        "predator_obs_range": 7, "prey_obs_range": 7})
        if args.workload == "drive":
            cfg["enable_drive_channels"] = True
    B = args.envs
    obs_dtype = torch.float64 if args.obs_dtype == "f64" else torch.float32
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    n_sub = max(1, args.streams)
    group = SubBatchedPredPreyGrass(cfg, batch_size=B, n_sub=n_sub, device=device, obs_dtype=obs_dtype, env_class=env_class,
                                    seed=args.seed + rank * B, **extra_kw, **({"_library": _emu_library()} if dry else {}))
    if args.workload == "walls":
        for e in group.subs:
            e.set_walls(cfg["manual_wall_positions"])
    group.reset()
    group.synchronize()
    env = group.subs[0]

    step_no = [0]

    def one_step():
        # every --rebalance-every steps the library re-sorts the env -> workgroup assignment by the envs' current number of
        # agents (ppg_rebalance: scheduling only, part of the timed loop)
        if args.rebalance_every > 0 and step_no[0] % args.rebalance_every == 0:
            group.rebalance()
        step_no[0] += 1
        group.step(random_actions=True, auto_reset=True)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(device)
    # zero the observation counters (bandwidth accounting) -- outside the timed region
    calls0 = []
    for e in group.subs:
        e.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1] = 0
        calls0.append(e.env_state[:, _abi.ENV_CALLS].clone())
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    # HIP events on the streams the kernels are launched on
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in group.streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in group.streams]
    t0 = time.perf_counter()
    for s, e in zip(group.streams, ev0):
        e.record(s)
    for _ in range(args.steps):
        one_step()
    for s, e in zip(group.streams, ev1):
        e.record(s)
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    # mean launch-to-launch time of the step kernel on each stream (the n_sub streams run concurrently)
    dev_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / len(ev0)
    if distributed:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # counters of the timed region, snapshotted before anything else steps the envs
    import numpy as np
    es = np.concatenate([e.env_state.cpu().numpy() for e in group.subs]).astype("int64")
    calls0 = np.concatenate([c.cpu().numpy() for c in calls0])
    assert ((es[:, _abi.ENV_CALLS] - calls0) == args.steps).all()

    # ---- optional leg: the RCCL observation all-gather north_star specifies (N > 1 only) ----
    gather_info = gather_overlapped = None
    if distributed and args.gather_steps > 0:
        try:
            from predpreygrass_amd.distributed import ObservationGatherer
            gs = [ObservationGatherer(e) for e in group.subs]
            torch.cuda.synchronize(device)
            dist.barrier()
            tg = time.perf_counter()
            nbytes = 0
            for _ in range(args.gather_steps):
                one_step()
                group.synchronize()
                for g in gs:
                    g.gather()
                    nbytes += g.last_bytes
            torch.cuda.synchronize(device)
            dist.barrier()
            tg = time.perf_counter() - tg
            t = torch.tensor([tg], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_info = {"value": round(n_gpus * B * args.gather_steps / float(t.item()), 1), "unit": "env-steps/s",
                           "steps": args.gather_steps, "ms_per_step": round(float(t.item()) / args.gather_steps * 1e3, 4),
                           "gathered_bytes_per_step_per_rank": int(nbytes / args.gather_steps),
                           "what": "step + synchronous all-gather (RCCL) of the compacted float64 observations, ids, rewards, flags"}
        except Exception as ex:  # never lose the main measurement to the optional leg
            gather_info = {"error": repr(ex)[:300]}
        # the same with the gather of step t overlapped with step t+1: the rows in use are packed (copied) on each
        # sub-batch's stream right after its step, the next step is enqueued behind the copy, and the collective runs on a
        # side stream that only waits for the copies
        try:
            from predpreygrass_amd.distributed import ObservationGatherer
            gs = [ObservationGatherer(e) for e in group.subs]
            cuda = not dry
            side = torch.cuda.Stream(device=device) if cuda else None
            torch.cuda.synchronize(device)
            dist.barrier()
            tg = time.perf_counter()
            nbytes = 0
            for _ in range(args.gather_steps):
                packed, events = [], []
                for g, s in zip(gs, group.streams):
                    if cuda:
                        with torch.cuda.stream(s):
                            packed.append(g.pack_local())
                            ev = torch.cuda.Event()
                            ev.record(s)
                            events.append(ev)
                    else:
                        packed.append(g.pack_local())
                one_step()                               # step t+1 runs while obs(t) travel
                if cuda:
                    for ev in events:
                        side.wait_event(ev)
                    with torch.cuda.stream(side):
                        for g, loc in zip(gs, packed):
                            for v in loc.values():
                                v.record_stream(side)   # allocated on the sub-batch's stream, consumed on the side stream
                            g.gather(loc)
                            nbytes += g.last_bytes
                else:
                    for g, loc in zip(gs, packed):
                        g.gather(loc)
                        nbytes += g.last_bytes
            torch.cuda.synchronize(device)
            dist.barrier()
            tg = time.perf_counter() - tg
            t = torch.tensor([tg], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_overlapped = {"value": round(n_gpus * B * args.gather_steps / float(t.item()), 1), "unit": "env-steps/s",
                                 "steps": args.gather_steps, "ms_per_step": round(float(t.item()) / args.gather_steps * 1e3, 4),
                                 "gathered_bytes_per_step_per_rank": int(nbytes / args.gather_steps),
                                 "what": "all-gather of step t's compacted observations overlapped with step t+1 (packed copies, side stream)"}
        except Exception as ex:
            gather_overlapped = {"error": repr(ex)[:300]}

    # ---- accounting ------------------------------------------------------------------
    status = int((es[:, _abi.ENV_STATUS]).max())
    n_obs_pred = int(es[:, _abi.ENV_OBS_PRED].sum())
    n_obs_prey = int(es[:, _abi.ENV_OBS_PREY].sum())
    G, Rp, Rq = env.grid_size, env.Rp, env.Rq
    osz = 8 if obs_dtype == torch.float64 else 4
    env_steps_rank = B * args.steps
    # 2*(3*G^2*8) + sum_obs(4*R^2*osz) + 64*L per env-step, L = rows in use
    cp, cq = env.obs_pred.shape[2], env.obs_prey.shape[2]   # observation channels (4; more in the drive / walls variants)
    alg_bytes = env_steps_rank * 2 * 3 * G * G * 8 + n_obs_pred * cp * Rp * Rp * osz + \
        n_obs_prey * cq * Rq * Rq * osz + 64 * (n_obs_pred + n_obs_prey)
    # bytes this implementation has to move at minimum (no dense grid exists in HBM): observations +
    # row tables r/w (27 B read, 35 B written per row) + grass table (8 B r/w, 2 B read) + env words
    min_bytes = n_obs_pred * cp * Rp * Rp * osz + n_obs_prey * cq * Rq * Rq * osz + \
        62 * (n_obs_pred + n_obs_prey) + env_steps_rank * (env.n_grass * 18 + 2 * 64 + 8)
    # HBM bytes per launch from rocprofv3 PMC passes of this very command (FETCH_SIZE x2 + WRITE_SIZE,
    # MI355X_MICROARCH.md HBM section); only quoted when the settings match the profiled run.
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01", "g_bench_default_summary.json")))
        if B == 4096 and n_sub == prof["concurrent_launches"] and obs_dtype == torch.float64 and not dry and args.workload == "base":
            traffic = int(prof["hbm_traffic_per_launch_bytes"]["total_corrected"])
    except Exception:
        traffic = None
    # which step kernel the library picked (predpreygrass_amd/csrc/ppg_host.h: ppg_use_multiwave): four waves per env while the
    # GPU is not full (<= 3072 envs in flight) or when LDS admits at most 4 envs per CU
    multiwave = (B <= 3072) or (160 * 1024 // max(env.lds_bytes, 1) <= 4) or args.workload in ("drive", "walls")
    if os.environ.get("PPG_MULTIWAVE") is not None:
        multiwave = os.environ["PPG_MULTIWAVE"] != "0"
    kernel_s = dev_ms / 1e3 / args.steps          # per launch; n_sub launches are in flight concurrently
    achieved = alg_bytes / args.steps / kernel_s / 1e9   # all n_sub concurrent launches together
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
            "data": "synthetic" if not dry else "DRY RUN ON CPU (emulated kernel) -- not a measurement",
            "config": {
                "workload": (f"{B} envs x {G}x{G} grid per GPU, default config ({cfg['n_initial_active_predator']} predators / "
                             f"{cfg['n_initial_active_prey']} prey / 100 grass, "
                             f"obs {Rp}x{Rp} / {Rq}x{Rq} {args.obs_dtype}), device-side uniform random actions, "
                             f"auto-reset, observations written every step (BASELINE.json configs[{3 if args.workload == 'c4' else 2}])"
                             + (f"; DRIVE-CONDITIONED variant ({cp} / {cq} observation channels incl. the per-agent drive planes), "
                                "NOT the BASELINE.json headline config" if args.workload == "drive" else "")) if not rq else
                            (f"WALLS variant of the second-generation env (walls_occlusion zigzag layout, mask + visibility channel + "
                             f"line-of-sight moves, obs {cp}x{Rp}x{Rp} / {cq}x{Rq}x{Rq} {args.obs_dtype}), {B} envs x {G}x{G} grid per GPU; "
                             "NOT the BASELINE.json headline config") if args.workload == "walls" else
                            (f"SECOND-GENERATION env (red_queen config_env_base: 12 predators / 10+10 prey of two types / 100 "
                             f"grass, obs {Rp}x{Rp} / {Rq}x{Rq} {args.obs_dtype}), {B} envs x {G}x{G} grid per GPU, device-side "
                             "random actions and reproduction uniforms, auto-reset; NOT the BASELINE.json headline config"),
                "envs_per_gpu": B,
                "parallelism": f"batch-sharded x{n_gpus}, no data-path collective",
                "sub_batches_per_gpu": n_sub,
                "mean_agents_per_env": round((n_obs_pred + n_obs_prey) / env_steps_rank, 2),
                "status_bits": status,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # the same with the bytes this design has to touch (no dense grid in HBM): the honest utilisation figure
                # when the SURVEY 8(d) model's dense-grid term dominates (64x64 grids: frac > 1)
                "frac_of_touched_bytes": round(min_bytes / args.steps / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": "profiles/r01/g_bench_default_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                  "bytes per launch)" if traffic else None,
                "achieved_from_pmc_traffic": round(traffic * n_sub / kernel_s / 1e9, 1) if traffic else None,
                "write_pattern_ceiling": "the same observation write pattern with no compute: 79.6 us per 4096-env "
                                         "launch = 4.85 TB/s (profiles/r01/c_store_pattern_ceiling.txt); linear fill 6.5 TB/s",
                "kernel": {"walls": "ppgw3_step_q2" if multiwave else "ppg3_step_q2",
                           "drive": "ppgw4_step_q2" if multiwave else "ppg4_step_q2",
                           "red_queen": "ppgw2_step_q2" if multiwave else "ppg2_step_q2"}.get(
                               args.workload, "ppgw_step_q2" if multiwave else "ppg_step_q2"),
                "kernel_ms": round(kernel_s * 1e3, 5),
                "concurrent_launches": n_sub,
                "algorithmic_bytes_per_launch": int(alg_bytes / args.steps / n_sub),
                "implementation_min_bytes_per_launch": int(min_bytes / args.steps / n_sub),
                "note": "kernel_ms = mean launch-to-launch time of the step kernel on its stream (HIP events on that "
                        "stream); concurrent_launches such kernels (one per sub-batch of envs_per_gpu/concurrent_"
                        "launches envs) overlap in time, so achieved = concurrent_launches x algorithmic_bytes_per_"
                        "launch / kernel_ms. Algorithmic bytes follow SURVEY 8(d); its dense-grid term is never moved "
                        "by this design -- implementation_min_bytes_per_launch is what the kernel must actually touch.",
            },
        }
        if gather_info is not None:
            out["obs_gather"] = gather_info
        if gather_overlapped is not None:
            out["obs_gather_overlapped"] = gather_overlapped
        if not args.no_cpu_baseline and n_gpus == 1 and not dry:
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, seconds=args.cpu_seconds, workload=args.workload)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
