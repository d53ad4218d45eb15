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
config (6 predators / 8 prey / 100 grass, obs 7x7 / 9x9), in its STEADY STATE: before the W warm-up and
K timed steps an untimed pre-roll steps the envs until the mean number of agent rows per env is
stationary ("after 200 warm-up"; right after a reset every env holds its 14 initial
agents and a step writes 2.7x fewer observation bytes than the workload defines).

N GPUs = N independent shards of 4096 envs (weak scaling, no data-path collective: envs never
interact).  `python bench.py --gpus N` without a torch.distributed launcher starts the N ranks itself.
For N>1 two extra legs measure the step followed by the ONE RCCL all-gather of the packed observation
image that north_star specifies ("obs_gather", "obs_gather_overlapped"; bandwidth-bound on xGMI, see
DESIGN.md).  Within one GPU the 4096 envs are stepped as --streams (default 3) independent sub-batches on
separate HIP streams.  Their observation tensors are allocated with ppg_alloc_spread (--obs-spread, default 32: physical
pages from a large stretch of device memory, which is what HBM wants for the step's scattered writes -- profiles/EXPERIMENTS.md, round 3), and
of --placement-candidates (default 3) such buffer sets the fastest is kept; both are allocation choices, results never depend
on them.

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

from bench_support import (device_state, gather_per_rank, gpu_uuid, measure_traffic, per_rank_summary, sampled_state,  # noqa: E402
                           self_launch, start_state_sampler)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "r06", "bench_driver_summary.json")


def cpu_baseline(cfg, seed0, seconds=12.0, threads=None, workload="base"):
    """The CPU oracle (C restatement of the reference step(), kind "port") timed on this host on a
    bounded sample of the same workload: one env per thread, random actions, auto-reset."""
    import concurrent.futures as cf
    if workload in ("red_queen", "walls"):
        from oracle.rq_oracle import RQOracleEnv as OracleEnv
    else:
        from oracle.ppg_oracle import OracleEnv
    threads = threads or os.cpu_count() or 1
    chunk = 2000

    def work(i):
        if workload == "walls":
            env = OracleEnv(cfg, walls=True)
            env.set_walls(cfg["manual_wall_positions"])
        else:
            env = OracleEnv(cfg)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            n += env.rollout_random(seed0 + i, chunk)
        return n, time.perf_counter() - t0

    # single thread first (the scalar port: a third of the time), then all cores (ctypes releases the GIL)
    full_seconds, seconds = seconds, seconds / 3.0
    n1, t1 = work(0)
    seconds = full_seconds
    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(work, range(threads)))
    tot = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {
        "value": round(tot / wall, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
        "single_thread_value": round(n1 / t1, 1),
        "sample": f"{threads} threads x 1 env each, {'red_queen config_env_base' if workload == 'red_queen' else 'walls_occlusion zigzag config' if workload == 'walls' else 'default config'}, "
                  f"Philox random actions + auto-reset, {seconds:.0f} s per thread ({tot} env-steps); "
                  f"oracle/{'rq_oracle.c' if workload in ('red_queen', 'walls') else 'ppg_oracle.c'} (-O2 -ffp-contract=off)",
        "reference_python_fixed": "reference Python step(): 45 env-steps/s on 1 core, ~400 on 8 cores "
                                  "(measured in the survey container, BASELINE.md section 2)",
    }


class HipBackend:
    """Where the bench runs: the MI355X of this rank.  (tests/bench_dry.py substitutes a CPU stand-in that drives the same
    control flow through the emulated kernel; nothing of it lives here.)"""
    dry = False
    dist_backend = "nccl"

    def setup(self, distributed, local_rank):
        import torch
        import torch.distributed as dist
        if local_rank >= torch.cuda.device_count():   # (device_count does not initialise the GPU)
            raise SystemExit(f"bench.py: rank with LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPUs are visible")
        if distributed:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = torch.device("cuda", local_rank if distributed else 0)
        torch.cuda.set_device(device)
        return device

    def env_kwargs(self):
        return {}

    def event(self):
        import torch
        return torch.cuda.Event(enable_timing=True)

    def synchronize(self, device):
        import torch
        torch.cuda.synchronize(device)

    def current_stream(self, device):
        import torch
        return torch.cuda.current_stream(device)

    def new_stream(self, device):
        import torch
        return torch.cuda.Stream(device=device)

    def stream_ctx(self, stream):
        import torch
        return torch.cuda.stream(stream)


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--obs-dtype", choices=["f64", "f32", "bf16"], default=None,
                    help="observation dtype (default: f64 for the base workload, f32 -- the reference's -- for red_queen, bf16 -- the "
                         "compact rows the policy kernels stage without conversion -- for policy_rollout)")
    ap.add_argument("--policy-arch", choices=["rllib", "fc256", "r3", "depth"], default="rllib",
                    help="policy_rollout: the network (rllib = what RLlib builds from the reference's model_config)")
    ap.add_argument("--workload", choices=["base", "c4", "red_queen", "drive", "walls", "policy_rollout", "dict_api"], default="base",
                    help="base: BASELINE.json configs[2] (the headline); c4: configs[3] (64x64 grid, 16 predators / 32 prey, "
                         "7x7 windows); red_queen: the second-generation env (SURVEY 8(f) N2) with its reference config; "
                         "drive: the drive-conditioned variant of the default config; walls: the walls variant with the "
                         "reference's zigzag layout and every line-of-sight option on; policy_rollout: the headline envs driven by "
                         "the two policy networks of the reference's PPO setup evaluated on the matrix cores next to the env "
                         "(SURVEY 8(f) N4; no observation leaves the GPU, roofline = bf16 MFMA); dict_api: the drop-in classes an existing "
                         "script imports -- PredPreyGrass(config).step(action_dict) for one env and VectorPredPreyGrass(64) -- driven from "
                         "the host like the reference's random_policy.py, every call crossing PCIe both ways (calls/s; latency-bound)")
    ap.add_argument("--policy-graph", action="store_true",
                    help="policy_rollout: the step (policy launch + ppg_step + the key's increment) captured once into a HIP graph and "
                         "replayed (predpreygrass_amd.policy.GraphedPolicyStep); the event leg stays un-graphed")
    ap.add_argument("--policy-open-loop", action="store_true",
                    help="policy_rollout, timing experiments only: the policy's actions go to a scratch tensor and the envs are stepped with "
                         "device-side random actions -- the population then does not depend on what the (possibly ablated) kernels compute")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0,
                    help="cpu_baseline: wall seconds of the all-cores leg (a single-thread leg of a third of that runs first)")
    ap.add_argument("--streams", type=int, default=None,
                    help="independent sub-batches per GPU, each on its own HIP stream (1 = one launch per step).  2 vs 3 with the "
                         "cooperative kernels, us per 4096-env step in one gpurun call each: GPU in its fast state 68.3 / 71.2, 67.2 / "
                         "66.4; in its slow state 87.7 / 83.6, 76.8 / 71.0 -- three are never far behind and clearly ahead when it counts.  "
                         "Default: 3; the walls and drive workloads 2 (their kernels are bound by per-env work, not by the write streams: "
                         "92.3 vs 94.9 and 180.1 vs 183.3 us per step, profiles/r05/l_walls_drive_pair_kernels_ab.txt)")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="after the timed region: a second leg of this many steps timed with HIP events on the launch streams "
                         "(roofline.kernel_ms_sustained); 0 = skip")
    ap.add_argument("--wave-plan", default=None,
                    help="A/B experiments: 'waves,helper_min_rows,coop_envs' forced on every sub-batch (ppg_set_wave_plan); the default is "
                         "the library's own choice, which the JSON line names (roofline.kernel)")
    ap.add_argument("--rebalance-every", type=int, default=64,
                    help="call ppg_rebalance every that many steps (0 = never): heavy envs are assigned to workgroups first")
    ap.add_argument("--preroll-min", type=int, default=3072,
                    help="untimed steps before --warmup that bring the envs into the steady state the workload is defined on "
                         "(all envs are reset together, so the population first overshoots to 41 rows/env, drops to 34 when the "
                         "survivors are truncated together at max_steps = 1000, and is within 1 %% of its long-run mean of 36.5 "
                         "from about step 2900 on: tools/exp_population.py)")
    ap.add_argument("--preroll-max", type=int, default=8192,
                    help="the pre-roll continues in 64-step windows until the mean rows per env of two consecutive windows "
                         "differ by < 1 %%, at most this many steps (0 = no pre-roll: measures the post-reset transient)")
    ap.add_argument("--fused-steps", type=int, default=100,
                    help="third leg (headline workload on a cooperative plan): ppg_rollout with this many transitions per launch, for as "
                         "many steps as the sustained leg (reported as `fused_rollout`, never as `value`); 0 = skip")
    ap.add_argument("--device-warm-seconds", type=float, default=2.0,
                    help="before the pre-roll the envs are stepped for this long (wall clock) and then reset again: a GPU that has been idle needs about a second of "
                         "load to reach its sustained clocks -- the first bench process on a fresh box measured 75 us per step where "
                         "every later one measured 66-68 (profiles/r03) -- and the workload is defined in its steady state")
    ap.add_argument("--obs-spread", type=int, default=32,
                    help="BatchedPredPreyGrass(obs_spread=N): the observation tensors live on 2 MB physical pages picked at random from N times "
                         "their size of device memory (ppg_alloc_spread, HIP virtual memory management): 62-64 us per 4096-env step at N = "
                         "32-64 where plain allocations draw from 62-91 us (profiles/r03/e_placement_experiments.txt).  N = 32 costs ~2.5 s and "
                         "58 GB of transient device memory at the headline size.  0 = torch's allocator")
    ap.add_argument("--placement-candidates", type=int, default=3,
                    help="SubBatchedPredPreyGrass(placement_candidates=K): where the driver puts the observation tensors in HBM decides "
                         "whether the step's scattered writes run in 62-66 or 76-81 us (same box, same process, same kernel: "
                         "profiles/r03/e_placement_experiments.txt); the constructor builds K candidate buffer sets side by side, steps each for "
                         "~70 ms and keeps the fastest (0.6 s and 14 GB of transient memory at K = 8).  1 = take the first allocation as it comes")
    ap.add_argument("--measure-traffic", dest="measure_traffic", action="store_true", default=None,
                    help="roofline.traffic from PMC passes of this command on THIS box (two child runs under rocprofv3 --pmc, ~25 s); "
                         "default: on for the single-GPU headline workload, off otherwise")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank, so that the gather legs run (needs torchrun)")
    ap.add_argument("--gather-steps", type=int, default=50,
                    help="N>1 only: extra legs of this many steps with the RCCL all-gather of the packed observation image")
    ap.add_argument("--gather-wire", choices=["f32", "native"], default="native",
                    help="observation dtype on the wire in the gather legs (f32 halves the bytes of float64 observations)")
    return ap.parse_args(argv)


def dict_api(args, backend, device):
    """`--workload dict_api`: what a script gets that only changes its import (BASE:219,473; one env per runner, TUNE:179-185).
    Leg 1: `PredPreyGrass(config)` -- reset(seed), then step(action_dict) with host-side random actions for every live agent (the protocol
    of random_policy.py / SURVEY Appendix B), reset again when the episode ends.  Leg 2: `VectorPredPreyGrass(64)`, one action dict per
    env, auto-reset.  Leg 3 (comparison): leg 1's loop with the data movement of rounds 1-4 -- one host->device copy of the actions, ten
    `.cpu()` copies of the tables and two of the observation slabs per call -- behind the same dict assembly."""
    import numpy as np
    import torch
    from predpreygrass_amd import _abi
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.env import PredPreyGrass, VectorPredPreyGrass
    dry = backend.dry
    kw = backend.env_kwargs()
    lib_kw = {"_library": kw["_library"]} if "_library" in kw else {}
    cfg = dict(config_env)
    rng = np.random.default_rng(args.seed)
    seconds = 0.3 if dry else max(2.0, min(args.cpu_seconds, 6.0))

    def drive_one(env, step_fn, budget):
        obs, _ = env.reset(seed=args.seed)
        live = list(obs)
        n, agents, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            actions = {a: int(v) for a, v in zip(live, rng.integers(0, 9, len(live)))}
            o, r, te, tr, _ = step_fn(env, actions)
            n += 1
            agents += len(o)
            live = [a for a in o if not te.get(a, False)]
            if te["__all__"] or tr["__all__"]:
                obs, _ = env.reset(seed=args.seed + n)
                live = list(obs)
        return n / (time.perf_counter() - t0), agents / max(n, 1)

    def legacy_step(env, action_dict):   # rounds 1-4: thirteen synchronous copies per call
        b, i = env._b, env._i
        rk, in_order = env._stage(action_dict)
        b.actions[i].copy_(torch.from_numpy(np.array(b.stage_actions(i))))
        b.step() if in_order else b.step(act_rank=rk[None].to(b.device))
        t = b.host_tables(i)
        es = t["env_state"][0]
        nP, nQ = max(int(es[_abi.ENV_N_PRED_ROWS]), 1), max(int(es[_abi.ENV_N_PREY_ROWS]), 1)
        obs = ({i: b.obs_pred[i, :nP].cpu().numpy()}, {i: b.obs_prey[i, :nQ].cpu().numpy()})
        t["row_info"] = np.zeros((1, b.S), dtype=np.uint8)
        return env._collect(False, {k: np.concatenate([v] * (i + 1)) for k, v in t.items()} if i else t, obs)

    one = PredPreyGrass(cfg, device=None if dry else device, **lib_kw)
    drive_one(one, lambda e, a: e.step(a), seconds / 4)   # warm-up
    rate_one, agents_one = drive_one(one, lambda e, a: e.step(a), seconds)
    rate_legacy, _ = drive_one(one, legacy_step, seconds / 2)
    nv = 4 if dry else 64
    vec = VectorPredPreyGrass(cfg, num_envs=nv, device=None if dry else device, seed=args.seed, auto_reset=True, **lib_kw)
    res = vec.reset(seed=args.seed)
    lives = [list(o) for o, _ in res]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        acts = [{a: int(v) for a, v in zip(lv, rng.integers(0, 9, len(lv)))} for lv in lives]
        out = vec.step(acts)
        lives = [[a for a in o if not te.get(a, False)] for o, r, te, tr, info in out]
        n += 1
    rate_vec = n * nv / (time.perf_counter() - t0)
    value = rate_vec
    us_one = 1e6 / rate_one
    # algorithmic bytes that cross PCIe per call of leg 1: the observation blocks in use + the env's state record + the action row
    bytes_call = agents_one * (0.15 * 4 * 49 * 8 + 0.85 * 4 * 81 * 8) + 9000 + one._b.S
    out = {
        "metric": "env-steps/sec at 4096x(25x25) grids, 1/2/4/8 MI355X; % HBM roofline",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": 1, "steps": n, "warmup": 0,
        "ms_per_step": round(1e3 / (rate_vec / nv), 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not dry else "DRY RUN ON CPU (emulated kernel) -- not a measurement",
        "config": {
            "workload": ("DICT API, NOT the BASELINE.json headline config: the reference's own interface (BASE:219,473) driven from the host -- "
                         f"`value` = VectorPredPreyGrass({nv}).step(list of action dicts) with auto-reset, default config, host-side random actions for "
                         "every live agent; every call = one host->device copy of the actions, one kernel launch, one gather launch, ONE device->host "
                         "copy (ppg_fetch), the reference's dicts rebuilt in Python"),
            "single_env": {"calls_per_s": round(rate_one, 1), "us_per_call": round(us_one, 1), "mean_agents_per_call": round(agents_one, 1),
                           "what": "PredPreyGrass(config).reset(seed) / .step(action_dict), one env (what one RLlib env runner holds, TUNE:179-185)"},
            "single_env_rounds_1_to_4_data_movement": {"calls_per_s": round(rate_legacy, 1), "us_per_call": round(1e6 / rate_legacy, 1),
                                                       "what": "the same loop with thirteen synchronous copies per call (ten table .cpu(), two observation slabs, one action row)"},
            "speedup_vs_rounds_1_to_4": round(rate_one / rate_legacy, 2),
            "vector_env": {"envs": nv, "env_steps_per_s": round(rate_vec, 1), "calls_per_s": round(rate_vec / nv, 1)},
            "reference_python_fixed": "reference Python step(): 45 env-steps/s (default config) / 441 (C1) on 1 core, ~400 on 8 cores (BASELINE.md section 2)",
        },
        "roofline": {"bound": "hbm", "achieved": round(bytes_call * rate_one / 1e9, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(bytes_call * rate_one / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                     "note": "a latency-bound path by construction (one env per call, PCIe round trip + Python dict assembly per call): the fraction "
                             "of the HBM peak is printed for the contract's sake and says nothing here; the HBM-bound figure is the default workload's"},
    }
    if not args.no_cpu_baseline and not dry:
        from oracle.ppg_oracle import OracleEnv
        env = OracleEnv(cfg)
        t0, steps = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(args.cpu_seconds, 4.0):
            steps += env.rollout_random(args.seed, 2000)
        out["cpu_baseline"] = {"value": round(steps / (time.perf_counter() - t0), 1), "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "oracle/ppg_oracle.c, one env on one host thread, default config, random actions + auto-reset "
                                         "(no Python dicts: the C restatement's own loop)"}
    print(json.dumps(out), flush=True)


MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)


def policy_rollout(args, backend, device, distributed, rank, n_gpus):
    """`--workload policy_rollout`: every step = ppg_policy_act (both species' networks of tune_ppo_base_environment.py:106-141,
    random-initialised weights, actions sampled from the softmax) + ppg_step with those actions, auto-reset.  Not the headline
    config: the same envs with the policy closed around them on the device.  Roofline: bf16 MFMA, real (unpadded) flops of
    the six layers over the policy kernels' time (HIP events around ppg_policy_act on the launch stream)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.policy import FusedPolicy, PolicyNet
    if backend.dry:
        raise SystemExit("policy_rollout needs the MFMA kernels: no dry run")
    cfg = dict(config_env)
    B = args.envs
    args.obs_dtype = args.obs_dtype or "bf16"
    obs_dtype = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[args.obs_dtype]
    env = BatchedPredPreyGrass(cfg, batch_size=B, device=device, obs_dtype=obs_dtype, seed=args.seed + rank * B)
    torch.manual_seed(1234)
    # --policy-arch rllib (default): what RLlib builds from tune_ppo_base_environment.py:106-141 -- channels-last reading, conv 3x3
    # 16/32/64, ONE Linear head (tests/golden/rllib_checkpoint/); fc256: the same encoder with head_fcnet_hiddens [256, 256];
    # r3: rounds 2-3's network (channel-first image, 256/256 head, channel-major flatten)
    arch_kw = {"rllib": dict(), "fc256": dict(head_hiddens=(256, 256)), "r3": dict(layout="chw", head_hiddens=(256, 256)), "depth": dict()}[args.policy_arch]
    if args.policy_arch == "depth":   # the newer tune scripts' rule (utils/networks.py: build_module_spec): (R - 1) // 2 convolutions of 16, 32, 64, 64 ...
        depth = lambda R: tuple(([16, 32, 64] + [64] * 8)[:(R - 1) // 2])
        nets = [PolicyNet(env.Rp, conv_channels=depth(env.Rp)), PolicyNet(env.Rq, conv_channels=depth(env.Rq))]
    else:
        nets = [PolicyNet(env.Rp, **arch_kw), PolicyNet(env.Rq, **arch_kw)]
    fused = FusedPolicy(nets[0], nets[1], device=device)
    arch_text = {"rllib": "channels-last 4 x R image with R channels, conv 3x3 16/32/64, flatten, ONE Linear(flat, 9) head: what RLlib's "
                          "DefaultPPOTorchRLModule builds from that model_config (fcnet_hiddens is ignored for image observations; pinned by the "
                          "reference tree's own checkpoint, tests/golden/rllib_checkpoint/)",
                 "fc256": "channels-last conv 3x3 16/32/64 + head_fcnet_hiddens [256, 256] + Linear(256, 9)",
                 "r3": "rounds 2-3's reading: channel-first R x R image, conv 3x3 16/32/64 + FC 256/256/9",
                 "depth": "channels-last, (R - 1) // 2 convolutions 16/32/64/64.. (the newer tune scripts' build_module_spec), ONE Linear head"}[args.policy_arch]
    env.reset()
    t_step = [0]
    scratch_actions = [torch.empty_like(env.actions)] if args.policy_open_loop else None

    def one_step(timed=None):
        if timed is not None:
            timed[0].record()
        fused.act(env, actions=scratch_actions, sample=True, seed=args.seed * 1000003 + t_step[0])
        if timed is not None:
            timed[1].record()
        t_step[0] += 1
        if args.policy_open_loop:
            env.step(random_actions=True, auto_reset=True)
        else:
            env.step(env.actions, auto_reset=True)

    preroll = min(args.preroll_min, 512) if args.preroll_max > 0 else 0    # (a forward pass costs ~10x a step: shorter pre-roll)
    for _ in range(preroll + args.warmup):
        one_step()
    graphed = None
    if args.policy_graph and not args.policy_open_loop:
        from predpreygrass_amd.policy import GraphedPolicyStep
        graphed = GraphedPolicyStep(fused, env, seed=args.seed * 1000003 + t_step[0])
        graphed.replay(args.warmup)
    torch.cuda.synchronize(device)
    env.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1] = 0
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    # the timed region: exactly --steps steps, NO events inside (an event record between two kernels of a stream costs ~5 us of idle
    # GPU on each side of the policy call: rocprofv3 timestamps, profiles/r04/v_policy_rollout_step_timeline.txt)
    t0 = time.perf_counter()
    if graphed is not None:
        graphed.replay(args.steps)
    else:
        for k in range(args.steps):
            one_step()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    wall = time.perf_counter() - t0
    per_rank = None
    if distributed:
        per_rank = gather_per_rank(dist, n_gpus, {"rank": rank, "ms_per_step": round(wall / args.steps * 1e3, 5), "gpu_uuid": gpu_uuid(device)})
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    es = env.env_state.cpu().numpy().astype("int64")
    n_pred, n_prey = int(es[:, _abi.ENV_OBS_PRED].sum()), int(es[:, _abi.ENV_OBS_PREY].sum())
    # second leg (untimed for `value`): the policy kernels' own time, HIP events around every policy call of up to 100 more steps
    ev_steps = min(args.steps, 100)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ev_steps)]
    for k in range(ev_steps):
        one_step(evs[k])
    torch.cuda.synchronize(device)
    pol_ms = sum(a.elapsed_time(b) for a, b in evs) / ev_steps * args.steps   # (scaled to the timed region's step count)
    es2 = env.env_state.cpu().numpy().astype("int64")
    ev_pred, ev_prey = int(es2[:, _abi.ENV_OBS_PRED].sum()) - n_pred, int(es2[:, _abi.ENV_OBS_PREY].sum()) - n_prey
    flops = 2.0 * (n_pred * fused.macs_per_observation(0) + n_prey * fused.macs_per_observation(1))
    ev_flops = 2.0 * (ev_pred * fused.macs_per_observation(0) + ev_prey * fused.macs_per_observation(1))
    achieved = ev_flops / (pol_ms / args.steps * ev_steps * 1e-3) / 1e12
    if rank == 0:
        out = {
            "metric": "env-steps/sec at 4096x(25x25) grids, 1/2/4/8 MI355X; % HBM roofline",
            "value": round(n_gpus * B * args.steps / wall, 1), "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": (f"POLICY ROLLOUT, NOT the BASELINE.json headline config: {B} envs x {env.grid_size}x{env.grid_size} grid per GPU, "
                             f"default config, every step = the two policy networks of the reference's PPO setup ({arch_text}; "
                             f"random-initialised, bf16 MFMA with fp32 accumulation) evaluated on the {args.obs_dtype} observation rows "
                             "in place, actions sampled on the device, then ppg_step with those actions and auto-reset"),
                "envs_per_gpu": B, "parallelism": f"batch-sharded x{n_gpus}, no data-path collective", "preroll_steps": preroll,
                "mean_agents_per_env": round((n_pred + n_prey) / (B * args.steps), 2),
                **({"hip_graph": "the timed region replays ONE captured step (policy launch, ppg_step, key increment): GraphedPolicyStep"} if graphed is not None else {}),
                **({"open_loop": "TIMING EXPERIMENT: the envs were stepped with random actions, the policy's actions discarded"} if args.policy_open_loop else {}),
                "bytes_per_agent_a_consumer_has_to_move": 1,
            },
            "roofline": {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": {"rllib": "ppg_policy_pipe2_16_8_" if os.environ.get("PPG_POLICY_FUSED", "1") != "0" else "ppg_policy_pipe{8,16}_", "depth": "ppg_policy_pipe8_ (predators) + ppg_policy_deep16_ (prey) "}.get(args.policy_arch, "ppg_policy_forward_") + args.obs_dtype,
                         "policy_arch": args.policy_arch,
                         "macs_per_observation": [fused.macs_per_observation(0), fused.macs_per_observation(1)],
                         "kernel_ms": round(pol_ms / args.steps, 5),
                         "kernel_ms_leg": f"a SECOND leg of {ev_steps} steps with HIP events around every policy call (an event record between two "
                                          "kernels costs ~5 us of idle GPU, so the timed region has none): kernel_ms, achieved and frac come "
                                          "from it and its own observation counts; value and ms_per_step from the event-free timed region",
                         "flops_per_step": int(flops / args.steps),
                         "flops_per_step_event_leg": int(ev_flops / ev_steps),
                         "note": "flops = 2 x multiply-accumulates of the network's layers (no padding counted) x observations evaluated; "
                                 "kernel_ms = the policy's launch(es) per step: ONE fused launch for both species (ppg_policy_pipe2_*; "
                                 "PPG_POLICY_FUSED=0: a plan launch + one forward launch per species)"},
        }
        out.update(per_rank_summary(per_rank, n_gpus, B))
        if not args.no_cpu_baseline and n_gpus == 1:
            n_threads = min(os.cpu_count() or 1, 32)   # (more threads than that make these small convolutions slower, not faster)
            torch.set_num_threads(n_threads)
            x = [torch.rand(2048, 4, env.Rp, env.Rp), torch.rand(2048, 4, env.Rq, env.Rq)]
            cost = []   # seconds per observation on the host, per species (bounded: --cpu-seconds in total)
            with torch.no_grad():
                for net, xx in zip(nets, x):
                    net = net.to("cpu")
                    net(xx)
                    net(xx)
                    tc, reps = time.perf_counter(), 0
                    while time.perf_counter() - tc < args.cpu_seconds / 2:
                        net(xx)
                        reps += 1
                    cost.append((time.perf_counter() - tc) / reps / xx.shape[0])
            tp, tq = cost
            mp, mq = n_pred / (B * args.steps), n_prey / (B * args.steps)
            out["cpu_baseline"] = {"value": round(1.0 / (mp * tp + mq * tq), 1), "unit": "env-steps/s", "cores": n_threads, "kind": "port",
                                   "sample": f"the same two networks as float32 PyTorch modules on the host ({n_threads} threads, batches of 2048): "
                                             f"{tp * 1e6:.1f} us per predator observation, {tq * 1e6:.1f} us per prey observation, x {mp:.1f} / {mq:.1f} "
                                             "observations per env-step; the env transition itself (0.5 M env-steps/s on these cores) is not included"}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def main(argv=None, backend=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    backend = backend or HipBackend()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args, argv))      # before anything touches the GPU
    world = int(env_world or "1")
    if env_world is not None and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # PMC passes of this command in child processes, before this process touches the GPU (single-GPU headline workload only)
    pmc = None
    want_pmc = args.measure_traffic if args.measure_traffic is not None else (args.workload == "base" and args.envs == 4096)
    if want_pmc and not args.traffic_child and world == 1 and not args.force_dist and not (backend and backend.dry):
        pmc = measure_traffic(argv, args, __file__)

    # clocks / power / who else is on the GPU while the loop runs: a side process, started before this one touches the GPU
    sampler = (None, "not started: dry run, a rank other than 0, a PMC child run or no sustained leg")
    # (not under rocprofv3: its preloaded tool library has initialised the GPU in THIS process before main() runs, so starting the
    #  side process here would already be an exec from a GPU-initialised process)
    profiler_attached = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "HSA_TOOLS")) for k in os.environ)
    if not (backend and backend.dry) and int(os.environ.get("RANK", "0")) == 0 and not args.traffic_child and args.sustained_steps > 0 \
            and args.workload not in ("policy_rollout", "dict_api"):
        sampler = (None, "not started: a profiler is attached to this process") if profiler_attached else start_state_sampler()

    import numpy as np
    import torch
    import torch.distributed as dist

    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist   # --force-dist: exercise the RCCL legs with a world of one rank (test hook)
    dry = backend.dry
    device = backend.setup(distributed, local_rank)
    n_gpus = world if distributed else 1

    if args.workload == "policy_rollout":
        return policy_rollout(args, backend, device, distributed, rank, n_gpus)
    if args.workload == "dict_api":
        if distributed:
            raise SystemExit("bench.py: --workload dict_api is a one-process leg")
        return dict_api(args, backend, device)
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
    obs_dtype = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[args.obs_dtype]
    n_sub = max(1, args.streams if args.streams is not None else (2 if args.workload in ("walls", "drive") else 3))
    setup = (lambda e: e.set_walls(cfg["manual_wall_positions"])) if args.workload == "walls" else None
    def build_group(spread):
        return SubBatchedPredPreyGrass(cfg, batch_size=B, n_sub=n_sub, device=device, obs_dtype=obs_dtype, env_class=env_class,
                                       seed=args.seed + rank * B, placement_candidates=1 if dry else args.placement_candidates,
                                       placement_setup=setup, **({} if dry else {"obs_spread": spread}),
                                       # (the headline workload's placements probe at 67-79 us per step where they are good and at 88-91
                                       #  where they are not: on a box that hands out mostly slow ones the search goes on for a while)
                                       placement_target_us=80.0 if (args.workload == "base" and B == 4096 and args.obs_dtype == "f64"
                                                                    and args.placement_candidates > 1) else None,
                                       # (at most ONE more than asked for: round 4's store_patterns5 runs showed that a GPU whose placements
                                       #  all probe slow stays slow whatever is allocated, and every candidate costs 2-3 s of untimed work)
                                       placement_max_candidates=args.placement_candidates + 1,
                                       **extra_kw,
                                       **backend.env_kwargs())

    spread_note = None
    try:
        group = build_group(args.obs_spread)
    except RuntimeError as exc:   # (the virtual-memory allocation is an optimisation: without it the run is slower, not wrong)
        if args.obs_spread <= 1 or "ppg_alloc_spread" not in str(exc):
            raise
        spread_note = f"ppg_alloc_spread failed ({exc}); observation tensors from torch's allocator"
        args.obs_spread = 0
        group = build_group(0)
    if args.wave_plan:
        wp = [int(v) for v in args.wave_plan.split(",")]
        for e in group.subs:
            e.set_wave_plan(*wp)
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

    def zero_obs_counters():
        for e in group.subs:
            e.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1] = 0

    def rows_written():
        return sum(int(e.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1].sum().item()) for e in group.subs)

    # ---- untimed pre-roll to the steady-state population -------------------------------------------
    WINDOW = 64
    preroll, trace = 0, []
    # device warm-up FIRST, then a fresh reset: the pre-roll below then runs the same number of steps in every run (unprofiled,
    # under the kernel trace, under the serialising PMC passes), so all of them time the same population
    warm_steps = 0
    if args.preroll_max > 0 and not dry and args.device_warm_seconds > 0:
        t_pre = time.perf_counter()
        while True:
            for _ in range(WINDOW):
                one_step()
            backend.synchronize(device)
            warm_steps += WINDOW
            warm = time.perf_counter() - t_pre >= args.device_warm_seconds
            if distributed:   # (the clocks differ: leave together)
                t = torch.tensor([0 if warm else 1], dtype=torch.int64, device=device)
                dist.all_reduce(t)
                warm = int(t.item()) == 0
            if warm:
                break
        group.reset()
        group.synchronize()
        step_no[0] = 0
    if args.preroll_max > 0:
        while True:   # (every exit decision below is agreed between the ranks)
            backend.synchronize(device)
            zero_obs_counters()
            for _ in range(WINDOW):
                one_step()
            backend.synchronize(device)
            preroll += WINDOW
            trace.append(rows_written() / (B * WINDOW))
            stationary = len(trace) >= 3 and all(abs(trace[-k] - trace[-k - 1]) <= 0.01 * trace[-k - 1] for k in (1, 2))
            if distributed:   # every rank has to leave the pre-roll after the same number of steps
                t = torch.tensor([0 if stationary else 1], dtype=torch.int64, device=device)
                dist.all_reduce(t)
                stationary = int(t.item()) == 0
            if (preroll >= args.preroll_min and stationary) or preroll >= args.preroll_max:
                break

    for _ in range(args.warmup):
        one_step()
    backend.synchronize(device)
    # zero the observation counters (bandwidth accounting) -- outside the timed region
    zero_obs_counters()
    calls0 = [e.env_state[:, _abi.ENV_CALLS].clone() for e in group.subs]
    backend.synchronize(device)
    if distributed:
        dist.barrier()
    backend.synchronize(device)
    # HIP events on the streams the kernels are launched on
    ev0 = [backend.event() for _ in group.streams]
    ev1 = [backend.event() for _ in group.streams]
    t0 = time.perf_counter()
    for s, e in zip(group.streams, ev0):
        e.record(s)
    for _ in range(args.steps):
        one_step()
    for s, e in zip(group.streams, ev1):
        e.record(s)
    backend.synchronize(device)
    if distributed:
        dist.barrier()
    backend.synchronize(device)
    wall = time.perf_counter() - t0
    # mean launch-to-launch time of the step kernel on each stream (the n_sub streams run concurrently)
    dev_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / len(ev0)
    per_rank = None
    if distributed:
        # `value` is set by the SLOWEST rank (max over ranks, below) -- and the GPUs of this pool differ by up to 25 % on this kernel with
        # identical clocks (profiles/r04/t_driver_command_on_eighteen_fresh_boxes.txt).  Every rank's own numbers travel with the line so
        # that a slow GPU can be told from a scaling problem.
        per_rank = gather_per_rank(dist, world, {
            "rank": rank, "ms_per_step": round(wall / args.steps * 1e3, 5), "kernel_ms": round(dev_ms / args.steps, 5),
            "placement_probe_us_min": None if not group.placement_probe_us else round(min(group.placement_probe_us), 1),
            "gpu_uuid": gpu_uuid(device, dry)})
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # counters of the timed region, snapshotted before anything else steps the envs
    es = np.concatenate([e.env_state.cpu().numpy() for e in group.subs]).astype("int64")
    calls0 = np.concatenate([c.cpu().numpy() for c in calls0])
    assert ((es[:, _abi.ENV_CALLS] - calls0) == args.steps).all()

    # ---- second leg: the same loop for --sustained-steps more steps, HIP events on the launch streams -------------------------
    # (the driver's 20-step window is 1.4 ms long: first-kernel latency and the drain of the last launches are 6-7 % of it; this leg
    # says what the same kernel does back to back.  Not `value`: the driver asked for exactly --steps steps.)
    sustained = None
    if args.sustained_steps > 0 and not dry:
        s0 = [backend.event() for _ in group.streams]
        s1 = [backend.event() for _ in group.streams]
        backend.synchronize(device)
        ts = time.perf_counter()
        for st, e in zip(group.streams, s0):
            e.record(st)
        for _ in range(args.sustained_steps):
            one_step()
        for st, e in zip(group.streams, s1):
            e.record(st)
        backend.synchronize(device)
        ts = time.perf_counter() - ts
        sustained = {"steps": args.sustained_steps,
                     "kernel_ms": sum(a.elapsed_time(b) for a, b in zip(s0, s1)) / len(s0) / args.sustained_steps,
                     "ms_per_step": ts / args.sustained_steps * 1e3}

    # clocks / power / temperatures WHILE the same loop runs (untimed extra steps on rank 0 for as long as one rocm-smi call takes)
    dev_state = {"unavailable": "not sampled: " + ("dry run" if dry else "not rank 0" if rank != 0 else "PMC child run" if args.traffic_child
                                                   else "--sustained-steps 0" if args.sustained_steps <= 0 else str(sampler[1]))}
    # (never under a profiler: its preloaded library initialises the GPU in every child process, and rocm-smi is a `#!/usr/bin/env python3`
    #  script -- an exec from a GPU-initialised process, which the GPU boxes refuse)
    under_profiler = bool(os.environ.get("LD_PRELOAD")) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if sampler[0] is not None:
        # untimed extra steps until the side process has taken one whole sample while they ran (at most 6 s)
        t_lo = time.time()
        got = None
        while got is None and time.time() - t_lo < 6.0:
            for _ in range(64):
                one_step()
            backend.synchronize(device)
            got = sampled_state(sampler[1], t_lo, time.time())
        try:
            sampler[0].kill()           # (this exact child)
            sampler[0].wait(timeout=5)
        except Exception:
            pass
        try:
            os.unlink(sampler[1])
        except OSError:
            pass
        if got is not None:
            try:
                got["gpu_uuid"] = str(torch.cuda.get_device_properties(torch.cuda.current_device()).uuid)
            except Exception:
                pass
            dev_state = got
        else:
            dev_state = {"unavailable": "the side sampler (rocm-smi in a process started before the GPU was touched) produced no whole sample in 6 s of stepping"}
    elif not dry and rank == 0 and not args.traffic_child and args.sustained_steps > 0 and not under_profiler:
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(state=device_state()))
        th.start()
        while th.is_alive():
            for _ in range(64):
                one_step()
            backend.synchronize(device)
        th.join()
        dev_state = box.get("state") or {"unavailable": "the sampling thread returned nothing"}

    # ---- third leg: the FUSED rollout -- ppg_rollout(n): n transitions per launch, the device-side random policy, observations
    # written every step exactly as above (bit-identical to n ppg_step calls; tests).  Reported next to `value`, never as `value`:
    # `value` is the per-step API, which also takes actions from outside.
    fused = None
    if args.fused_steps > 0 and not dry and args.workload in ("base", "red_queen") and group.subs[0].wave_plan()[2] > 0 \
            and group.subs[0].wave_plan()[0] == 4:
        n_f = args.fused_steps
        launches = max(1, args.sustained_steps // n_f) if args.sustained_steps > 0 else 4
        f0 = [backend.event() for _ in group.streams]
        f1 = [backend.event() for _ in group.streams]
        group.rollout(n_f, random_actions=True, auto_reset=True)      # (untimed: first launch of this kernel)
        backend.synchronize(device)
        zero_obs_counters()
        backend.synchronize(device)
        tf = time.perf_counter()
        for st, e in zip(group.streams, f0):
            e.record(st)
        for _ in range(launches):
            group.rollout(n_f, random_actions=True, auto_reset=True)
        for st, e in zip(group.streams, f1):
            e.record(st)
        backend.synchronize(device)
        tf = time.perf_counter() - tf
        rows_f = rows_written()
        G_, Rp_, Rq_ = env.grid_size, env.Rp, env.Rq
        es_f = np.concatenate([e.env_state.cpu().numpy() for e in group.subs]).astype("int64")
        npred_f, nprey_f = int(es_f[:, _abi.ENV_OBS_PRED].sum()), int(es_f[:, _abi.ENV_OBS_PREY].sum())
        osz_f = {torch.float64: 8, torch.float32: 4, torch.bfloat16: 2}[obs_dtype]
        bytes_f = (npred_f * 4 * Rp_ * Rp_ + nprey_f * 4 * Rq_ * Rq_) * osz_f + 62 * rows_f + B * launches * n_f * (env.n_grass * 18 + 2 * 64 + 8)
        ms_f = sum(a.elapsed_time(b) for a, b in zip(f0, f1)) / len(f0)
        fused = {"what": f"ppg_rollout({n_f}) x {launches} per sub-batch: {n_f} transitions per launch (ppgc_rollout / ppgc2_rollout: the workgroups of a launch run on "
                         "from step to step, no launch boundary), device-side random policy, auto-reset, observations written every step; "
                         "bit-identical to the same number of ppg_step calls (tests/test_hip_parity.py)",
                 "steps": launches * n_f, "value": round(n_gpus * B * launches * n_f / tf, 1), "unit": "env-steps/s",
                 "ms_per_step": round(tf / (launches * n_f) * 1e3, 5), "kernel_ms_per_step": round(ms_f / (launches * n_f), 5),
                 "mean_agents_per_env": round(rows_f / (B * launches * n_f), 2),
                 "bytes_written_per_second_over_hbm_peak": round(bytes_f / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "note": "NOT an HBM roofline figure and it may exceed 1: only the resident workgroups' envs are being stepped at any time "
                         "(1024 workgroups x 2 envs x 94 KB of observations = 193 MB, inside the 256 MB Infinity Cache), no kernel boundary "
                         "writes the caches back between steps, and every observation line is overwritten one step (~50 us) later -- much "
                         "of it never reaches HBM.  What this leg shows is the step without launch boundaries; the HBM-bound number is "
                         "`value` (per-step launches, every step's observations written back)."}

    # ---- optional legs: the ONE RCCL all-gather per step north_star specifies (N > 1 only) -------------------------
    gather_info = gather_overlapped = None
    extra_legs = {}
    if distributed and args.gather_steps > 0:
        from predpreygrass_amd.distributed import ObservationGatherer
        wire = torch.float32 if args.gather_wire == "f32" else None
        rows_now = (es[:, _abi.ENV_N_PRED_ROWS].mean(), es[:, _abi.ENV_N_PREY_ROWS].mean())
        # image sized from the current population with 30 % head room (every rank computes its own and the maximum is used)
        want = torch.tensor([rows_now[0] * 1.3 + 1, rows_now[1] * 1.3 + 1], dtype=torch.float64, device=device)
        dist.all_reduce(want, op=dist.ReduceOp.MAX)
        rows_per_env = tuple(float(v) for v in want.tolist())

        def leg(overlapped, mode="all_gather", include_obs=True):
            g = ObservationGatherer(group.subs, wire_dtype=wire, rows_per_env=rows_per_env, mode=mode, include_obs=include_obs)
            # untimed: one gather, then the image is sized from what it really used (max over the ranks + 15 %)
            one_step()
            if not dry:
                group.wait(backend.current_stream(device))
            if g.fit(g.gather()):
                g.gather()
            cuda = not dry
            cur = backend.current_stream(device) if cuda else None
            side = backend.new_stream(device) if (cuda and overlapped) else None
            backend.synchronize(device)
            dist.barrier()
            tg = time.perf_counter()
            for _ in range(args.gather_steps):
                one_step()
                if cuda:
                    group.wait(cur)                 # the pack reads what the sub-batch streams have just written
                slot = g.pack(stream=cur)
                if side is not None:
                    # the collective of step t runs on a side stream while step t+1 (which only waits for the pack) computes
                    side.wait_stream(cur)
                    with backend.stream_ctx(side):
                        g.gather(slot, pack=False)
                else:
                    g.gather(slot, pack=False)
            backend.synchronize(device)
            dist.barrier()
            tg = time.perf_counter() - tg
            leg_ranks = gather_per_rank(dist, world, {"rank": rank, "ms_per_step": round(tg / args.gather_steps * 1e3, 4)})
            t = torch.tensor([tg], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hs = g.headers()
            return {"value": round(n_gpus * B * args.gather_steps / float(t.item()), 1), "unit": "env-steps/s",
                    "steps": args.gather_steps, "ms_per_step": round(float(t.item()) / args.gather_steps * 1e3, 4),
                    "per_rank_ms_per_step": [r["ms_per_step"] for r in leg_ranks],
                    "collectives_per_step": 1,
                    "wire_bytes_per_step_per_rank": int(g.capacity),
                    "image_bytes_used_last_step": [int(h.bytes_used) for h in hs],
                    "image_overflows": int(sum(h.overflow for h in hs)),
                    "wire_obs_dtype": "f32" if (wire is not None or obs_dtype == torch.float32) else "f64",
                    "what": (("one all_gather_into_tensor (RCCL) of the packed observation image per step" if mode == "all_gather" else
                              "the all-gather of the packed observation image spelled as ONE grouped batch of point-to-point copies per step "
                              "(every image straight to every peer: the direct algorithm on the xGMI mesh)" if mode == "all_pairs" else
                              "one gather of the packed observation image to rank 0 per step (every other rank sends once, receives nothing)") +
                             ("" if include_obs else " WITHOUT the observation sections (PPG_PACK_NO_OBS: env words, ids, rewards, flags -- "
                              "what leaves a GPU when the policy runs next to the env)") +
                             (", the collective of step t overlapped with step t+1 (side stream, two image slots)" if overlapped
                              else ", synchronous: the next step waits for it"))}
        for name, kw in (("sync", dict(overlapped=False)), ("overlapped", dict(overlapped=True)),
                         ("obs_gather_to_root", dict(overlapped=False, mode="gather")),
                         ("obs_all_pairs", dict(overlapped=False, mode="all_pairs")),
                         ("ids_rewards_gather", dict(overlapped=False, include_obs=False))):
            try:   # never lose the main measurement to an optional leg
                res = leg(**kw)
            except Exception as ex:
                res = {"error": repr(ex)[:300]}
            if name == "overlapped":
                gather_overlapped = res
            elif name == "sync":
                gather_info = res
            else:
                extra_legs[name] = res

    # ---- accounting ------------------------------------------------------------------
    status = int(np.bitwise_or.reduce(es[:, _abi.ENV_STATUS]))
    # envs that have gone through the reference's own non-deterministic branch (BASE:759-764: all four neighbour cells taken, the
    # child lands on a free cell chosen by an unseeded global RNG) at least once since their last ppg_reset: there the build's
    # Philox choice is its own contract, everything else is the reference's arithmetic
    fallback_envs = int(((es[:, _abi.ENV_STATUS] & _abi.STATUS_FALLBACK_SPAWN) != 0).sum())
    n_obs_pred = int(es[:, _abi.ENV_OBS_PRED].sum())
    n_obs_prey = int(es[:, _abi.ENV_OBS_PREY].sum())
    G, Rp, Rq = env.grid_size, env.Rp, env.Rq
    osz = {torch.float64: 8, torch.float32: 4, torch.bfloat16: 2}[obs_dtype]
    env_steps_rank = B * args.steps
    cp, cq = env.obs_pred.shape[2], env.obs_prey.shape[2]   # observation channels (4; more in the drive / walls variants)
    obs_bytes = n_obs_pred * cp * Rp * Rp * osz + n_obs_prey * cq * Rq * Rq * osz
    # Bytes THIS RUN moved, counted by the kernel (rows in use per env and call): observations + row tables r/w (27 B read,
    # 35 B written per row) + grass table (8 B r/w + 2 B read per patch) + env words.  No dense grid exists in HBM.
    run_bytes = obs_bytes + 62 * (n_obs_pred + n_obs_prey) + env_steps_rank * (env.n_grass * 18 + 2 * 64 + 8)
    # SURVEY.md section 8(d)'s model of the REFERENCE's data structures: 2*(3*G^2*8) + sum_obs(4*R^2*osz) + 64*L per env-step
    # (its first term is a dense-grid read+write this design never performs) -- kept for comparison only
    survey_bytes = env_steps_rank * 2 * 3 * G * G * 8 + obs_bytes + 64 * (n_obs_pred + n_obs_prey)
    kernel_name = env._lib.ppg_step_kernel_name(env._handle).decode()   # what ppg_step launches for these handles (csrc/ppg_host.h: ppg_wave_plan)
    kernel_s = dev_ms / 1e3 / args.steps          # per launch; n_sub launches are in flight concurrently
    achieved = run_bytes / args.steps / kernel_s / 1e9   # all n_sub concurrent launches together
    value = n_gpus * env_steps_rank / wall
    # HBM bytes per launch from rocprofv3 PMC passes of the driver's command (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md
    # HBM section), scaled by this run's own counted bytes relative to the profiled run's (same workload, same pre-roll)
    traffic = traffic_src = traffic_origin = None   # traffic_origin: "in_run" (PMC passes of THIS run) | "profiles_fallback" | None
    if pmc is not None:
        traffic_origin = "in_run"
        scale = (run_bytes / args.steps / n_sub) / pmc["counted"]
        traffic = int(pmc["total"] * scale)
        traffic_src = (f"THIS run's box: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE passes (counters only, one child process each) of "
                       f"this command; means over the timed dispatches of {pmc['kernel']}: WRITE_SIZE {pmc['WRITE_SIZE']:.0f} B + 2 x FETCH_SIZE "
                       f"{pmc['FETCH_SIZE']:.0f} B = {pmc['total']:.0f} B per launch (the child counted {pmc['counted']} B), scaled x{scale:.4f} "
                       "by this run's own counted bytes")
    try:
        if pmc is not None:
            raise LookupError   # measured on this box: no need for the committed profile
        prof = json.load(open(PROFILE_SUMMARY))
        same = (prof["envs_per_gpu"] == B and prof["concurrent_launches"] == n_sub and prof["obs_dtype"] == args.obs_dtype
                and prof["workload"] == args.workload and not dry)
        if same:
            scale = (run_bytes / args.steps / n_sub) / prof["counted_bytes_per_launch"]
            traffic = int(prof["hbm_traffic_per_launch_bytes"]["total_corrected"] * scale)
            traffic_origin = "profiles_fallback"
            traffic_src = (f"profiles/r06/bench_driver_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                           f"{prof['hbm_traffic_per_launch_bytes']['total_corrected']} B per launch at "
                           f"{prof['mean_agents_per_env']} agents/env, scaled x{scale:.4f} by this run's counted bytes")
    except Exception:
        if pmc is None:
            traffic = None

    if rank == 0:
        roof = {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "bytes_counted": "observations + row tables + grass table + env words actually moved, from the kernel's own row counters "
                             "of the timed region",
            # SURVEY 8(d)'s formula adds a dense-grid read+write (30 KB per env-step at 25x25, 197 KB at 64x64) that this
            # design never moves; with it the figure can exceed 1 (64x64 grids), so it is a comparison number, not `frac`
            "frac_survey_formula": round(survey_bytes / args.steps / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_origin": traffic_origin,
            "traffic_source": traffic_src,
            "write_pattern_ceiling": "the same write pattern and launch structure with no compute (tools/store_patterns4.hip, "
                                     "profiles/r03) in plain hipMalloc memory: 61-73 us per 4096-env step depending on where the driver put "
                                     "the buffer; a linear fill of the observation tensors: 6.3 TB/s = 61 us for the bytes of a step -- which is what "
                                     "the step reaches on spread pages (config.obs_spread)",
            "kernel": kernel_name,
            "kernel_ms": round(kernel_s * 1e3, 5),
            "kernel_ms_sustained": None if sustained is None else round(sustained["kernel_ms"], 5),
            "frac_sustained": None if sustained is None else round(run_bytes / args.steps / (sustained["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "sustained_steps": None if sustained is None else sustained["steps"],
            "value_sustained": None if sustained is None else round(n_gpus * B / (sustained["ms_per_step"] * 1e-3), 1),
            "concurrent_launches": n_sub,
            "counted_bytes_per_launch": int(run_bytes / args.steps / n_sub),
            "survey_formula_bytes_per_launch": int(survey_bytes / args.steps / n_sub),
            "note": "kernel_ms = mean launch-to-launch time of the step kernel on its stream (HIP events on that "
                    "stream); concurrent_launches such kernels (one per sub-batch of envs_per_gpu/concurrent_"
                    "launches envs) overlap in time, so achieved = concurrent_launches x counted_bytes_per_"
                    "launch / kernel_ms.  kernel_ms_sustained / frac_sustained: the same over a second leg of sustained_steps "
                    "steps run right after the timed region (same envs, same loop).",
        }
        if not dry and achieved > HBM_PEAK_GBS:
            raise SystemExit(f"bench.py: achieved {achieved:.0f} GB/s exceeds the HBM peak -- the timing or the byte count is broken")
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
            "dtype": args.obs_dtype,
            "data": "synthetic" if not dry else "DRY RUN ON CPU (emulated kernel) -- not a measurement",
            "config": {
                "workload": (f"{B} envs x {G}x{G} grid per GPU, default config ({cfg['n_initial_active_predator']} predators / "
                             f"{cfg['n_initial_active_prey']} prey / 100 grass, "
                             f"obs {Rp}x{Rp} / {Rq}x{Rq} {args.obs_dtype}), device-side uniform random actions, "
                             f"auto-reset, observations written every step, steady-state population after an untimed pre-roll "
                             f"(BASELINE.json configs[{3 if args.workload == 'c4' else 2}])"
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
                "preroll_steps": preroll, "device_warm_steps": warm_steps,
                "device_state_under_load": dev_state,
                # the same as ONE scalar string (a consumer that flattens nested values keeps it)
                "device_state_summary": ("unavailable: " + str(dev_state["unavailable"])[:120]) if "unavailable" in dev_state else
                                        " ".join(f"{k}={dev_state[k]}" for k in ("sclk_MHz", "mclk_MHz", "power_W", "kfd_processes_listed", "gpu_uuid")
                                                 if k in dev_state),
                "obs_spread": args.obs_spread, **({"obs_spread_note": spread_note} if spread_note else {}),
                "placement_candidates_us_per_step": None if group.placement_probe_us is None else [round(v, 1) for v in group.placement_probe_us],
                # which placement mode this run drew (60 ms probes of each candidate buffer set; good placements of the headline workload
                # probe at 67-79 us per step, slow ones -- and every placement on a slow GPU of the pool -- at 80-91)
                "placement_probe_us": None if not group.placement_probe_us else {
                    "min": round(min(group.placement_probe_us), 1), "median": round(float(np.median(group.placement_probe_us)), 1),
                    "n": len(group.placement_probe_us)},
                "preroll_mean_agents_per_env_by_64_step_window": [round(v, 2) for v in trace[-8:]],
                "mean_agents_per_env": round((n_obs_pred + n_obs_prey) / env_steps_rank, 2),
                "status_bits": status,
                "fallback_spawn_envs": fallback_envs,
                "step_kernel": kernel_name,
                "wave_plan": list(env.wave_plan()),
            },
            "roofline": roof,
        }
        out.update(per_rank_summary(per_rank, n_gpus, B))
        if fused is not None:
            out["fused_rollout"] = fused
        if gather_info is not None:
            out["obs_gather"] = gather_info
        if gather_overlapped is not None:
            out["obs_gather_overlapped"] = gather_overlapped
        if gather_info is not None:
            out.update(extra_legs)
        if not args.no_cpu_baseline and not dry:   # rank 0's host cores, also next to the N > 1 curve (north_star)
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, seconds=args.cpu_seconds, workload=args.workload)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
