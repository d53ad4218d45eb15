"""The second-generation HIP kernel source, compiled for the CPU wave emulator (tests/wave_emu), against
(1) the golden vectors produced by the reference's red_queen env and (2) the pinned gen-2 oracle on random
rollouts.  No GPU needed; the GPU tests (test_hip_parity_rq.py) repeat this on the real kernels."""
import numpy as np
import pytest
import torch

from oracle.rq_oracle import RQOracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.red_queen import BatchedRedQueen
from tests import emu_backend
from tests.golden_io_rq import RQGoldenCase, case_names
from tests.parity_utils_rq import replay_golden_case, rollout_vs_oracle

CASES = case_names() + case_names(walls=True)


def make_env(cfg, B, **kw):
    return BatchedRedQueen(cfg, batch_size=B, _library=emu_backend.library(), **kw)


@pytest.mark.parametrize("name", CASES)
def test_emulated_kernel_replays_reference(name):
    env, n_ordered = replay_golden_case(make_env, name, max_calls=120)
    if "shuffled" in name:
        assert n_ordered > 10  # the explicit-order kernel variant was exercised


def test_emulated_random_rollout_matches_oracle():
    cfg = RQGoldenCase("rq_mixed_types_seed7").config
    env = make_env(cfg, 3)
    n_resets, stats = rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=41, n_calls=150, check_every=1, check_grid=True)
    assert n_resets >= 3 and stats["births"] > 5 and stats["type2"] > 0


def test_emulated_random_rollout_base_config_f64_obs():
    cfg = RQGoldenCase("rq_base_seed3").config
    env = make_env(cfg, 2, obs_dtype=torch.float64)
    n_resets, stats = rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=7, n_calls=60, check_every=3)
    assert stats["births"] > 0


def test_gen2_rejects_bad_configs():
    cfg = dict(RQGoldenCase("rq_trunc_seed4").config)
    with pytest.raises(ValueError):
        make_env(dict(cfg, type_1_action_range=4), 1)
    with pytest.raises(ValueError):
        make_env(dict(cfg, n_possible_type_1_predators=70000), 1)
    with pytest.raises(ValueError):
        BatchedRedQueen(None, _library=emu_backend.library())


def _walls_oracle(cfg, walls):
    o = RQOracleEnv(cfg, walls=True)
    o.set_walls(walls)
    return o


def test_emulated_random_rollout_with_walls_matches_oracle():
    """Device reset around the walls (Philox over the free cells), wall / corner / line-of-sight rules, masked
    observations with the visibility channel: every call against the oracle."""
    case = RQGoldenCase("wo_los_two_types_seed5")
    cfg, walls = case.config, case.wall_xy
    env = make_env(cfg, 3, walls=True)
    env.set_walls(walls)
    n_resets, stats = rollout_vs_oracle(env, lambda: _walls_oracle(cfg, walls), seed0=11, n_calls=120, check_every=1,
                                        check_grid=True)
    assert n_resets >= 3 and stats["births"] > 5
    info = env.row_info.numpy()
    assert (info <= 5).all()


def test_second_generation_kernels_are_clean_under_ubsan():
    """The second-generation, walls and drive code paths in the UBSan build of the emulated kernel (traps on signed
    overflow, bad shifts, misaligned access -- e.g. the odd-sized observation blocks of 5- and 7-channel variants)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from tests.emu_backend import library\n"
        "from tests.parity_utils_rq import replay_golden_case, rollout_vs_oracle\n"
        "from tests.parity_utils import rollout_vs_oracle as rollout_base\n"
        "from tests.golden_io_rq import RQGoldenCase\n"
        "from predpreygrass_amd.red_queen import BatchedRedQueen\n"
        "from predpreygrass_amd.batched import BatchedPredPreyGrass\n"
        "from predpreygrass_amd.config import config_env\n"
        "from oracle.rq_oracle import RQOracleEnv\n"
        "from oracle.ppg_oracle import OracleEnv\n"
        "lib = library(sanitize=True)\n"
        "mk = lambda cfg, B, **kw: BatchedRedQueen(cfg, batch_size=B, _library=lib, **kw)\n"
        "replay_golden_case(mk, 'rq_mixed_types_seed7')\n"
        "replay_golden_case(mk, 'rq_shuffled_seed5')\n"
        "replay_golden_case(mk, 'wo_los_two_types_seed5')\n"
        "replay_golden_case(mk, 'wo_mask_only_shuffled_seed6')\n"
        "cfg = RQGoldenCase('rq_pool_exhaust_seed2').config\n"
        "rollout_vs_oracle(mk(cfg, 2), lambda: RQOracleEnv(cfg), seed0=3, n_calls=120)\n"
        "drv = {**config_env, 'enable_drive_channels': True, 'grid_size': 11, 'initial_num_grass': 30, 'max_steps': 50}\n"
        "rollout_base(BatchedPredPreyGrass(drv, batch_size=2, _library=lib), lambda: OracleEnv(drv), seed0=5, n_calls=80)\n"
        "print('UBSAN-CLEAN')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "UBSAN-CLEAN" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


RQ_STATE = ("row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "row_lastrep", "env_state",
            "grass_xy", "grass_energy", "obs_pred", "obs_prey")


def _same_tables(a, b):
    for n in RQ_STATE:
        ta, tb = getattr(a, n), getattr(b, n)
        assert ta.cpu().numpy().tobytes() == tb.cpu().numpy().tobytes(), n


@pytest.mark.parametrize("coop,B", [(2, 5), (4, 3)])
def test_cooperative_fused_rollout_equals_single_steps(coop, B):
    """ppg_rollout on a second-generation handle = the fused form of its cooperative kernel (ppgc2_rollout): random policy with
    reproduction (device Philox uniforms), truncations and auto-resets inside the launch, then an action tape -- every table,
    status word and observation equal to the same number of ppg_step calls."""
    cfg = dict(RQGoldenCase("rq_mixed_types_seed7").config, max_steps=35)
    a, b = make_env(cfg, B, seed=13), make_env(cfg, B, seed=13)
    b.set_wave_plan(4, 0, coop)
    assert b.wave_plan() == (4, 0, coop)
    a.reset()
    b.reset()
    for _ in range(80):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(45, random_actions=True, auto_reset=True)
    b.rollout(35, random_actions=True, auto_reset=True)
    _same_tables(a, b)
    K = 30
    a, b = make_env(cfg, B, seed=3), make_env(cfg, B, seed=3)
    b.set_wave_plan(4, 0, coop)
    a.reset()
    b.reset()
    n_act = int(max(cfg.get("type_1_action_range", 3), cfg.get("type_2_action_range", 3))) ** 2
    tape = torch.randint(-1, min(n_act, 9), (K, B, a.S), generator=torch.Generator().manual_seed(2), dtype=torch.int8)
    for t in range(K):
        a.step(tape[t].contiguous())
    b.rollout(K, actions=tape)
    _same_tables(a, b)


def test_fused_rollout_needs_a_cooperative_plan_and_no_walls():
    cfg = RQGoldenCase("rq_base_seed3").config
    env = make_env(cfg, 2)
    env.set_wave_plan(1, 0, 0)
    env.reset()
    with pytest.raises(RuntimeError, match="cooperative"):
        env.rollout(3, random_actions=True)
    case = RQGoldenCase("wo_los_two_types_seed5")
    wenv = make_env(case.config, 2, walls=True)
    wenv.set_walls(case.wall_xy)
    wenv.reset()
    with pytest.raises(RuntimeError, match="cooperative"):
        wenv.rollout(3, random_actions=True)


def test_identity_act_rank_equals_no_rank_with_more_than_64_acting_prey():
    """the explicit-order path's prey row list (publish_order) reaches past byte 256 of the LDS scratch once more
    than 64 prey act; the move-cost table must live behind it.  110 prey, capacity 128, move cost on: identity ranks through
    ppg_step_ordered give what the plain step gives."""
    cfg = dict(RQGoldenCase("rq_base_seed3").config, move_energy_cost_factor=0.01, n_initial_active_type_1_prey=110,
               n_possible_type_1_prey=400, n_initial_active_type_2_prey=0, grid_size=25)
    a, b = make_env(cfg, 2, seed=5, prey_capacity=128), make_env(cfg, 2, seed=5, prey_capacity=128)
    a.reset()
    b.reset()
    gen = torch.Generator().manual_seed(4)
    for t in range(6):
        act = torch.randint(0, 9, (2, a.S), generator=gen, dtype=torch.int8)
        rows = a.host_tables()
        n_pred, n_prey = rows["env_state"][:, _abi.ENV_N_PRED_ROWS], rows["env_state"][:, _abi.ENV_N_PREY_ROWS]
        assert int(n_prey.min()) > 64
        rank = torch.zeros((2, a.S), dtype=torch.uint8)
        for e in range(2):   # identity: rank = position among the acting (= live) rows of the type, in row order
            for lo, n in ((0, int(n_pred[e])), (a.pred_capacity, int(n_prey[e]))):
                live = (rows["row_flags"][e, lo: lo + n] & _abi.ROW_DIED) == 0
                rank[e, lo: lo + n] = torch.from_numpy((np.cumsum(live) - 1).clip(0).astype(np.uint8))
        a.step(act)
        b.step(act, act_rank=rank)
        _same_tables(a, b)
