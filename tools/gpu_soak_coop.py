#!/usr/bin/env python3
"""Soak run on a GPU: the cooperative step kernels (every wave plan the library ships), with the observation tensors on spread
physical pages, long random rollouts with device reset / Philox actions / auto-reset, EVERY env compared with its CPU oracle on
EVERY call; then ppg_rollout against the same number of steps.   usage: gpu_soak_coop.py [envs=96] [calls=400] [walls]
(walls: only the walls variant's kernels, cooperative and not)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.ppg_oracle import OracleEnv  # noqa: E402
from oracle.rq_oracle import RQOracleEnv  # noqa: E402
from predpreygrass_amd.batched import BatchedPredPreyGrass  # noqa: E402
from predpreygrass_amd.config import config_env  # noqa: E402
from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base  # noqa: E402
from tests import parity_utils as P1, parity_utils_rq as P2  # noqa: E402
from tests.golden_io_rq import RQGoldenCase  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t0 = time.time()


def soak_walls():
    """round 6: the cooperative walls kernel (ppgc3_step) and the four-wave / pair kernels, all writing the listed rows' window cells as one
    run (Env::obs_cells_walls); every env with walls of its own, every golden configuration of the walls env, float32 and float64 rows."""
    import numpy as np
    from tests.golden_io_rq import case_names
    rng = np.random.default_rng(2)
    for name in case_names(walls=True):
        case = RQGoldenCase(name)
        cfg, G = dict(case.config, max_steps=120), case.config["grid_size"]
        base = np.asarray(case.wall_xy).reshape(-1, 2)
        walls = [np.unique(np.concatenate([base[b % 4::4], rng.integers(0, G, size=(3 + b % 9, 2))]), axis=0) for b in range(B)]
        for plan, dt in (((4, 0, 2), torch.float32), ((4, 0, 4), torch.float64), ((4, 0, 0), torch.float32), ((2, 0, 0), torch.float64)):
            env = BatchedRedQueen(cfg, batch_size=B, device="cuda:0", walls=True, obs_dtype=dt)
            env.set_wave_plan(*plan)
            assert env.wave_plan() == plan, (plan, env.wave_plan())
            env.set_walls(walls, per_env=True)
            made = []

            def oracle():
                o = RQOracleEnv(cfg, walls=True)
                o.set_walls(walls[len(made)])
                made.append(o)
                return o
            r = P2.rollout_vs_oracle(env, oracle, seed0=500 + plan[2], n_calls=calls, check_every=1, check_grid=True)
            print("walls", name, plan, env.step_kernel_name(), str(dt)[6:], "resets/stats", r, f"{time.time() - t0:.0f} s", flush=True)
            env.close()


if len(sys.argv) > 3 and sys.argv[3] == "walls":
    soak_walls()
    print("soak walls ok", f"{time.time() - t0:.0f} s")
    sys.exit(0)
short = {**config_env, "max_steps": 150}
for plan in ((4, 0, 2), (4, 0, 4), (6, 0, 3), (8, 0, 2), (16, 0, 1)):
    for cfg, dt in ((short, torch.float64), ({**short, "grid_size": 12, "initial_num_grass": 50}, torch.float32)):
        env = BatchedPredPreyGrass(cfg, batch_size=B, device="cuda:0", obs_dtype=dt, obs_spread=4)
        env.set_wave_plan(*plan)
        assert env.wave_plan() == plan, (plan, env.wave_plan())
        r = P1.rollout_vs_oracle(env, lambda cfg=cfg: OracleEnv(cfg), seed0=1000 + plan[0] * 10 + plan[2], n_calls=calls, check_grid=True)
        print("base", plan, env.step_kernel_name(), cfg["grid_size"], str(dt)[6:], "resets", r, f"{time.time() - t0:.0f} s", flush=True)
        env.close()
# round 6: the cooperative kernels WITHOUT a channel-0 cell map (ppgcm_*): what 64x64 grids get by default, and forced on small grids
c4 = {**short, "grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32, "predator_obs_range": 7, "prey_obs_range": 7}
for maps, cfg, dt in ((None, c4, torch.float64), ("3", short, torch.float64), ("3", {**short, "grid_size": 12, "initial_num_grass": 50}, torch.float32),
                      ("3", {**short, "grid_size": 8, "n_initial_active_predator": 6, "n_initial_active_prey": 18, "initial_num_grass": 30,
                             "energy_gain_per_step_grass": 0.5}, torch.float64)):   # (crowded: spawn fallbacks, never a full grid)
    if maps:
        os.environ["PPG_COOP_MAPS"] = maps
    for plan in ((4, 0, 2), (4, 0, 4)):
        env = BatchedPredPreyGrass(cfg, batch_size=B, device="cuda:0", obs_dtype=dt, obs_spread=4)
        env.set_wave_plan(*plan)
        assert env.step_kernel_name().startswith("ppgcm_step"), env.step_kernel_name()
        r = P1.rollout_vs_oracle(env, lambda cfg=cfg: OracleEnv(cfg), seed0=3000 + plan[2], n_calls=calls, check_grid=True)
        fb = int(env.env_state[:, 12].sum()) if hasattr(env, "env_state") else -1
        print("base three maps", plan, env.step_kernel_name(), cfg["grid_size"], str(dt)[6:], "resets", r, f"{time.time() - t0:.0f} s", flush=True)
        env.close()
    os.environ.pop("PPG_COOP_MAPS", None)
mixed = dict(RQGoldenCase("rq_mixed_types_seed7").config, max_steps=120)
os.environ["PPG_COOP_MAPS"] = "3"
env = BatchedRedQueen(mixed, batch_size=B, device="cuda:0", obs_spread=4)
env.set_wave_plan(4, 0, 2)
assert env.step_kernel_name() == "ppgcm2_step_q2", env.step_kernel_name()
r = P2.rollout_vs_oracle(env, lambda: RQOracleEnv(mixed), seed0=91, n_calls=calls, check_every=1, check_grid=True)
print("gen2 three maps", env.step_kernel_name(), "resets/stats", r, f"{time.time() - t0:.0f} s", flush=True)
env.close()
os.environ.pop("PPG_COOP_MAPS", None)
for plan in ((4, 0, 2), (4, 0, 4)):
    for cfg in (dict(config_env_base, max_steps=150), mixed):
        env = BatchedRedQueen(cfg, batch_size=B, device="cuda:0", obs_spread=4)
        env.set_wave_plan(*plan)
        r = P2.rollout_vs_oracle(env, lambda cfg=cfg: RQOracleEnv(cfg), seed0=77 + plan[2], n_calls=calls, check_every=1, check_grid=True)
        print("gen2", plan, env.step_kernel_name(), "resets/stats", r, f"{time.time() - t0:.0f} s", flush=True)
        env.close()
# fused rollouts at full size: 4096 envs, cooperative plan, spread pages, against per-step launches on plain tensors
for cls, cfg, name in ((BatchedPredPreyGrass, {**config_env, "max_steps": 200}, "base"), (BatchedRedQueen, dict(config_env_base, max_steps=200), "gen2")):
    a = cls(cfg, batch_size=4096, device="cuda:0", seed=5)
    b = cls(cfg, batch_size=4096, device="cuda:0", seed=5, obs_spread=8)
    b.set_wave_plan(4, 0, 2)
    a.reset()
    b.reset()
    for _ in range(600):
        a.step(random_actions=True, auto_reset=True)
    for _ in range(6):
        b.rollout(100, random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    for n in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(a, n), getattr(b, n)), (name, n)
    print(name, "4096 envs: 600 steps == 6 x ppg_rollout(100) on spread pages", f"{time.time() - t0:.0f} s", flush=True)
soak_walls()
print("soak ok", f"{time.time() - t0:.0f} s")
