"""The MULTI-WAVE step kernels (ppgw*_step_*: wave 0 steps the env, all four / eight wavefronts of the workgroup write
the survivors' observations after one workgroup barrier) under the CPU wave emulator: nwaves x 64 fibers, wave
collectives per wave, wg_barrier() across the waves that are still running, random resume order across lanes AND
waves.  Same golden vectors and oracles as the single-wave emulator tests; the GPU tests compare the real kernels
(test_hip_parity.py::test_multiwave_step_kernels_give_identical_results)."""
import pytest
import torch

from oracle.ppg_oracle import OracleEnv
from oracle.rq_oracle import RQOracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
from predpreygrass_amd.red_queen import BatchedRedQueen
from tests import emu_backend
from tests.golden_io_rq import RQGoldenCase
from tests.parity_utils import replay_golden_cases, rollout_vs_oracle
from tests.parity_utils_rq import replay_golden_case
from tests.parity_utils_rq import rollout_vs_oracle as rq_rollout_vs_oracle


def make_base(cfg, B, **kw):
    return BatchedPredPreyGrass(cfg, batch_size=B, _library=emu_backend.library(), **kw)


def make_gen2(cfg, B, **kw):
    return BatchedRedQueen(cfg, batch_size=B, _library=emu_backend.library(), **kw)


@pytest.mark.parametrize("waves,names,max_calls", [
    ("4", ["default_seed0"], 150),              # descriptors in registers (7x7 / 9x9 windows)
    ("8", ["default_seed1"], 150),
    ("4", ["c4_seed0"], 100),                   # 64x64 grid
    ("2", ["c4_seed0"], 100),                   # the pair kernel (what the library picks for 64x64 grids: 8-bit maps, 7 envs per CU)
    ("2", ["dense_seed3"], None),               # ghost cells / co-occupancy through the pair kernel
    ("8", ["even_obs_seed0"], None),            # even windows, ends with the truncation call
    ("4", ["kickback_fast_seed5"], 120),        # reward kick-backs to grandparents
    ("4", ["drive_default_seed2"], 100),        # drive variant: four-wave kernel with per-wave window staging
    ("8", ["drive_custom_lists_big_windows_seed4"], 60),   # (8 asks for the variant's widest multi-wave kernel: 4)
    ("2", ["drive_default_seed2"], 60),         # drive variant: the pair kernel
])
def test_base_family_golden_cases_multiwave(waves, names, max_calls, monkeypatch):
    monkeypatch.setenv("PPG_EMU_WAVES", waves)
    replay_golden_cases(make_base, names, config_env, max_calls=max_calls)


def test_wave_count_takes_effect(monkeypatch):
    lib = emu_backend.library()
    for waves, drive, expect in (("1", False, 1), ("2", False, 2), ("4", False, 4), ("8", False, 8), ("8", True, 4), ("2", True, 2)):
        monkeypatch.setenv("PPG_EMU_WAVES", waves)
        env = make_base({**config_env, "enable_drive_channels": drive}, 1)
        env.reset(seed=1)
        env.step(random_actions=True)
        assert lib.ppg_emu_last_waves() == expect


@pytest.mark.parametrize("waves,name", [
    ("4", "rq_mixed_types_seed7"),
    ("8", "rq_base_seed3"),
    ("4", "wo_zigzag_seed1"),                   # walls: line-of-sight staging area per wave
    ("4", "wo_los_two_types_seed5"),
    ("2", "wo_zigzag_seed1"),                   # walls: the pair kernel
    ("2", "wo_los_two_types_seed5"),
])
def test_second_generation_golden_cases_multiwave(waves, name, monkeypatch):
    monkeypatch.setenv("PPG_EMU_WAVES", waves)
    replay_golden_case(make_gen2, name, max_calls=100)


@pytest.mark.parametrize("waves", ["2", "4", "8"])
def test_multiwave_random_rollout_matches_oracle_and_single_wave(waves, monkeypatch):
    """Device reset + Philox actions + auto-reset on a config with the generic (LDS descriptor) observation path and
    more prey than one register holds; every call against the oracle, and the final state against the single-wave run."""
    cfg = {**config_env, "grid_size": 11, "n_initial_active_predator": 9, "n_initial_active_prey": 30, "initial_num_grass": 40,
           "predator_obs_range": 11, "prey_obs_range": 13, "max_steps": 50, "energy_gain_per_step_grass": 0.3}
    states = []
    for w in ("1", waves):
        monkeypatch.setenv("PPG_EMU_WAVES", w)
        env = make_base(cfg, 2, prey_capacity=128)
        rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=31, n_calls=70, check_grid=True)
        states.append({n: getattr(env, n).clone() for n in
                       ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey")})
        states[-1]["env_state"] = env.env_state[:, : _abi.ENV_CALLS].clone()
    for n, t in states[0].items():
        assert torch.equal(t, states[1][n]), n


def test_multiwave_second_generation_rollout_matches_oracle(monkeypatch):
    monkeypatch.setenv("PPG_EMU_WAVES", "8")
    cfg = RQGoldenCase("rq_mixed_types_seed7").config
    env = make_gen2(cfg, 2)
    n_resets, stats = rq_rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=43, n_calls=100, check_every=1, check_grid=True)
    assert stats["births"] > 3


@pytest.mark.parametrize("waves,family", [("4", "base"), ("2", "base"), ("4", "gen2")])
def test_helper_waves_only_for_heavy_envs(waves, family, monkeypatch):
    """KParams::helper_min_rows: the helper wavefronts of envs with fewer agent rows exit at once and wave 0 writes every row; envs
    cross the threshold in both directions during the rollout (population 14 .. 60+).  Every call against the oracle."""
    monkeypatch.setenv("PPG_EMU_WAVES", waves)
    monkeypatch.setenv("PPG_HELPER_MIN_ROWS", "16" if family == "base" else "30")
    if family == "base":
        env = make_base(dict(config_env), 3)
        rollout_vs_oracle(env, lambda: OracleEnv(dict(config_env)), seed0=17, n_calls=90, check_grid=True)
        rows = env.env_state[:, _abi.ENV_N_PRED_ROWS] + env.env_state[:, _abi.ENV_N_PREY_ROWS]
        assert int(rows.max()) >= 16 > int(rows.min())   # (both sides of the threshold were exercised; the rollout starts at 14 rows)
    else:
        cfg = RQGoldenCase("rq_mixed_types_seed7").config
        env = make_gen2(cfg, 2)
        rq_rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=43, n_calls=80, check_every=1, check_grid=True)


def test_unknown_wave_count_is_rejected(monkeypatch):
    monkeypatch.setenv("PPG_EMU_WAVES", "3")
    env = make_base(dict(config_env), 1)
    env.reset(seed=1)
    with pytest.raises(RuntimeError):
        env.step(random_actions=True)
