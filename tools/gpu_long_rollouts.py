#!/usr/bin/env python3
"""Long random rollouts on the GPU, every env compared with its oracle on every call (device reset, Philox actions /
reproduction uniforms, auto-reset): base default config, second-generation base config, walls config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ppg_oracle import OracleEnv
from oracle.rq_oracle import RQOracleEnv
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
from tests.golden_io_rq import RQGoldenCase
from tests import parity_utils as P1, parity_utils_rq as P2

B, calls = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
env = BatchedPredPreyGrass(config_env, batch_size=B, device="cuda:0")
print("base resets", P1.rollout_vs_oracle(env, lambda: OracleEnv(config_env), seed0=123456, n_calls=calls, check_grid=True), f"{time.time()-t0:.0f} s", flush=True)
drv = {**config_env, "enable_drive_channels": True}
env = BatchedPredPreyGrass(drv, batch_size=B, device="cuda:0")
print("drive resets", P1.rollout_vs_oracle(env, lambda: OracleEnv(drv), seed0=223456, n_calls=calls // 2), f"{time.time()-t0:.0f} s", flush=True)
env = BatchedRedQueen(config_env_base, batch_size=B, device="cuda:0")
print("gen2", P2.rollout_vs_oracle(env, lambda: RQOracleEnv(config_env_base), seed0=654321, n_calls=calls, check_grid=True), f"{time.time()-t0:.0f} s", flush=True)
c = RQGoldenCase("wo_los_two_types_seed5")
def orc():
    o = RQOracleEnv(c.config, walls=True); o.set_walls(c.wall_xy); return o
env = BatchedRedQueen(c.config, batch_size=B, device="cuda:0", walls=True); env.set_walls(c.wall_xy)
print("walls", P2.rollout_vs_oracle(env, orc, seed0=777, n_calls=calls, check_grid=True), f"{time.time()-t0:.0f} s", flush=True)
