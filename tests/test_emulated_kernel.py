"""The kernel SOURCE (predpreygrass_amd/csrc/ppg_kernel.h) compiled for the CPU wave emulator
(tests/wave_emu) must reproduce the reference bit-for-bit.  This is a CPU-side check of the device
code and of the host wrappers; the GPU parity tests proper are in test_hip_parity.py (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle.ppg_oracle import OracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.batched import BatchedPredPreyGrass, lexkey
from predpreygrass_amd.config import config_env
from tests.emu_backend import library
from tests.parity_utils import replay_golden_cases, rollout_vs_oracle


def make_env(cfg, B, **kw):
    return BatchedPredPreyGrass(cfg, batch_size=B, _library=library(), **kw)


@pytest.mark.parametrize("names,max_calls", [
    (["c1_seed0"], None),
    (["default_seed0", "default_seed1"], 260),
    (["c4_seed0"], None),
    (["dense_seed0", "dense_seed3"], None),
    (["rewards_seed3"], None),
    (["pool_seed3"], None),
    (["even_obs_seed0"], None),
    (["seasonal_short_seed0"], None),      # generated from base_environment_seasonal's own file
    (["seasonal_default_seed1"], None),
    (["plus_eating_seed2"], None),         # ...sparse_rewards_plus_eating's own file
    (["dense_rewards_seed0"], None),       # ...base_environment_dense_rewards' own file (reward = energy delta)
    (["dense_additive_seed4"], None),      # ...base_environment_dense_rewards_additive's own file
    (["kickback_seed0"], None),            # ...base_environment_sparse_rewards_plus_kickback's own file
    (["kickback_fast_seed5"], None),       # 194 grandparent rewards incl. double kicks and kick-after-own-reproduction
    (["drive_default_seed2"], None),       # ...drive_conditioned_environment's own file (3 + 4 drive channels)
    (["drive_custom_lists_big_windows_seed4"], None),   # 13x13 / 15x15 windows (numpy's pairwise sum splits), custom lists
])
def test_golden_cases_through_emulated_kernel(names, max_calls):
    replay_golden_cases(make_env, names, config_env, max_calls=max_calls)


def test_truncation_call_after_max_steps():
    """E8: max_steps real steps, then one all-truncated call that does not advance current_step."""
    replay_golden_cases(make_env, ["even_obs_seed0"], config_env)  # 200 steps + truncation call


@pytest.mark.parametrize("over,B,calls,cap", [
    ({}, 3, 120, 128),
    ({"n_initial_active_predator": 4, "n_initial_active_prey": 8, "initial_num_grass": 30}, 3, 150, 64),
    ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20, "initial_num_grass": 25,
      "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 150}, 4, 200, 128),
    ({"grid_size": 5, "n_initial_active_predator": 5, "n_initial_active_prey": 9, "initial_num_grass": 8,
      "predator_obs_range": 3, "prey_obs_range": 5, "max_steps": 60, "energy_gain_per_step_grass": 0.5}, 4, 200, 256),
])
def test_random_rollout_matches_oracle(over, B, calls, cap):
    """Device-side Philox reset + random actions + auto-reset vs oracle ppo_rollout_random."""
    cfg = {**config_env, **over}
    env = make_env(cfg, B, prey_capacity=cap)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=99, n_calls=calls, check_grid=True)


def test_float32_observations_are_rounded_float64():
    cfg = dict(config_env)
    env = make_env(cfg, 2, obs_dtype=__import__("torch").float32)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=5, n_calls=40)


def test_lexkey_orders_like_python_string_sort():
    ids = list(range(0, 1300)) + [1999, 2000, 9999, 10000, 54321, 99999, 100000, 999999]
    keys = lexkey(ids)
    by_key = [i for _, i in sorted(zip(keys.tolist(), ids))]
    assert by_key == sorted(ids, key=lambda i: str(i))
    lib = library()
    assert [int(lib.ppg_lexkey(i)) for i in ids] == keys.tolist()


def test_prey_row_overflow_is_flagged_not_silent():
    from predpreygrass_amd import _abi
    cfg = {**config_env, "energy_gain_per_step_grass": 0.3, "initial_num_grass": 200, "max_steps": 300}
    env = make_env(cfg, 1, prey_capacity=64)
    env.reset(seed=3)
    for _ in range(120):
        env.step(random_actions=True)
    st = int(env.env_state[0, _abi.ENV_STATUS])
    assert st & _abi.STATUS_PREY_OVERFLOW
    assert int(env.env_state[0, _abi.ENV_N_PREY_ROWS]) <= 64


def test_invalid_configs_are_rejected():
    with pytest.raises(ValueError, match="Cannot place more unique positions"):
        make_env({**config_env, "grid_size": 5, "initial_num_grass": 30}, 1)
    with pytest.raises(ValueError):
        make_env({**config_env, "grid_size": 200}, 1)


def test_kernel_source_is_clean_under_ubsan():
    """The kernel source built with -fsanitize=undefined -fno-sanitize-recover (CPU wave-emulator build):
    any signed overflow, out-of-range shift, misaligned or out-of-bounds constant index aborts the process."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from tests.emu_backend import library\n"
        "from tests.parity_utils import replay_golden_cases, rollout_vs_oracle\n"
        "from predpreygrass_amd.batched import BatchedPredPreyGrass\n"
        "from predpreygrass_amd.config import config_env\n"
        "from oracle.ppg_oracle import OracleEnv\n"
        "lib = library(sanitize=True)\n"
        "mk = lambda cfg, B: BatchedPredPreyGrass(cfg, batch_size=B, _library=lib)\n"
        "replay_golden_cases(mk, ['dense_seed0', 'dense_seed3'], config_env)\n"
        "replay_golden_cases(mk, ['c4_seed0'], config_env)\n"
        "cfg = {**config_env, 'grid_size': 5, 'n_initial_active_predator': 5, 'n_initial_active_prey': 9,\n"
        "       'initial_num_grass': 8, 'predator_obs_range': 3, 'prey_obs_range': 5, 'max_steps': 60,\n"
        "       'energy_gain_per_step_grass': 0.5}\n"
        "rollout_vs_oracle(mk(cfg, 3), lambda: OracleEnv(cfg), seed0=9, n_calls=150)\n"
        "import os\n"
        "for w in ('4', '8'):\n"                      # the multi-wave step kernels (helper-wave code paths)
        "    os.environ['PPG_EMU_WAVES'] = w\n"
        "    for name in ['default_seed0', 'drive_default_seed2'][:1 if w == '8' else 2]:\n"
        "        replay_golden_cases(mk, [name], config_env, max_calls=80)\n"
        "    rollout_vs_oracle(mk(cfg, 2), lambda: OracleEnv(cfg), seed0=10, n_calls=80)\n"
        "print('UBSAN-CLEAN')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "UBSAN-CLEAN" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_kernel_source_is_clean_under_address_sanitizer():
    """The kernel source built with -fsanitize=address (CPU wave-emulator build), the bytes behind every workgroup's LDS poisoned
    (wave_emu.h): out-of-bounds reads and writes of LDS, of the "device" tensors (torch's CPU allocations, intercepted through the
    preloaded runtime) and of the host code's own buffers abort the process.  Single-wave, multi-wave and cooperative kernels (both env-region
    layouts), the second generation, the cooperative walls kernel, the pack and fetch launches."""
    import subprocess, sys, os
    from tests.emu_backend import asan_runtime, build
    rt = asan_runtime()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc has no libasan.so here")
    build(sanitize="address")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "from tests.emu_backend import library\n"
        "from tests.parity_utils import replay_golden_cases, rollout_vs_oracle\n"
        "from predpreygrass_amd.batched import BatchedPredPreyGrass\n"
        "from predpreygrass_amd.config import config_env\n"
        "from oracle.ppg_oracle import OracleEnv\n"
        "lib = library(sanitize='address')\n"
        "mk = lambda cfg, B: BatchedPredPreyGrass(cfg, batch_size=B, _library=lib)\n"
        "replay_golden_cases(mk, ['dense_seed0'], config_env)\n"
        "replay_golden_cases(mk, ['c4_seed0'], config_env)\n"
        "cfg = {**config_env, 'grid_size': 5, 'n_initial_active_predator': 5, 'n_initial_active_prey': 9,\n"
        "       'initial_num_grass': 8, 'predator_obs_range': 3, 'prey_obs_range': 5, 'max_steps': 60,\n"
        "       'energy_gain_per_step_grass': 0.5}\n"
        "rollout_vs_oracle(mk(cfg, 3), lambda: OracleEnv(cfg), seed0=9, n_calls=100)\n"
        "for w in ('4', '8'):\n"
        "    os.environ['PPG_EMU_WAVES'] = w\n"
        "    replay_golden_cases(mk, ['default_seed0'], config_env, max_calls=60)\n"
        "    rollout_vs_oracle(mk(cfg, 2), lambda: OracleEnv(cfg), seed0=10, n_calls=60)\n"
        "os.environ.pop('PPG_EMU_WAVES')\n"
        "def coop(c, B):\n"                       # the cooperative kernels: 1 KB pieces through the padded maps
        "    e = mk(c, B); e.set_wave_plan(4, 0, 2); return e\n"
        "replay_golden_cases(coop, ['default_seed0'], config_env, max_calls=60)\n"
        "cfg7 = {**config_env, 'grid_size': 9, 'predator_obs_range': 13, 'prey_obs_range': 15, 'initial_num_grass': 20, 'max_steps': 40}\n"
        "rollout_vs_oracle(coop(cfg7, 3), lambda: OracleEnv(cfg7), seed0=4, n_calls=50)\n"
        "os.environ['PPG_COOP_MAPS'] = '3'\n"    # the cooperative kernels WITHOUT a channel-0 map and without halos (ppgcm_*): every window
        "replay_golden_cases(coop, ['default_seed0'], config_env, max_calls=60)\n"   # element outside the grid must stay inside LDS
        "replay_golden_cases(coop, ['c4_seed0'], config_env, max_calls=40)\n"
        "rollout_vs_oracle(coop(cfg, 3), lambda: OracleEnv(cfg), seed0=12, n_calls=80)\n"
        "os.environ.pop('PPG_COOP_MAPS')\n"
        "e = mk(config_env, 3); e.reset(seed=1)\n"
        "for _ in range(5):\n"
        "    e.step(random_actions=True, auto_reset=True); e.fetch(); e.fetch(1, 1)\n"
        "from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base\n"
        "from oracle.rq_oracle import RQOracleEnv\n"
        "from tests.parity_utils_rq import rollout_vs_oracle as rq_rollout\n"
        "rq_rollout(BatchedRedQueen(config_env_base, batch_size=2, _library=lib), lambda: RQOracleEnv(config_env_base), seed0=3, n_calls=40)\n"
        "from tests.golden_io_rq import RQGoldenCase\n"   # the cooperative walls kernel (ppgc3_step): whole rows, bitmaps and staging areas of other envs' regions
        "for name in ('wo_los_two_types_seed5', 'wo_mask_only_shuffled_seed6'):\n"
        "    case = RQGoldenCase(name)\n"
        "    w = BatchedRedQueen(case.config, batch_size=3, walls=True, _library=lib); w.set_wave_plan(4, 0, 2); w.set_walls(case.wall_xy)\n"
        "    assert w.step_kernel_name().startswith('ppgc3_step'), w.step_kernel_name()\n"
        "    def wo(case=case):\n"
        "        o = RQOracleEnv(case.config, walls=True); o.set_walls(case.wall_xy); return o\n"
        "    rq_rollout(w, wo, seed0=6, n_calls=40)\n"
        "print('ASAN-CLEAN')\n" % root)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", PYTHONMALLOC="malloc")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800, env=env)
    assert out.returncode == 0 and "ASAN-CLEAN" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def _state(env):
    names = ["row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "grass_xy",
             "grass_energy", "obs_pred", "obs_prey"]
    st = {n: getattr(env, n).clone() for n in names}
    st["env_state"] = env.env_state.clone()
    return st


def _assert_same_state(a, b, env):
    import torch
    from predpreygrass_amd import _abi
    assert torch.equal(a["env_state"], b["env_state"])
    nP, nQ = a["env_state"][:, _abi.ENV_N_PRED_ROWS], a["env_state"][:, _abi.ENV_N_PREY_ROWS]
    cp = env.pred_capacity
    rows = torch.arange(env.S)[None, :]
    used = (rows < nP[:, None]) | ((rows >= cp) & (rows < cp + nQ[:, None]))
    for n in ("row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward"):
        assert torch.equal(a[n][used], b[n][used]), n
    assert torch.equal(a["grass_xy"], b["grass_xy"]) and torch.equal(a["grass_energy"], b["grass_energy"])
    assert torch.equal(a["obs_pred"][used[:, :cp]], b["obs_pred"][used[:, :cp]])
    assert torch.equal(a["obs_prey"][used[:, cp:]], b["obs_prey"][used[:, cp:]])


@pytest.mark.parametrize("over,K", [
    ({}, 60),
    ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20, "initial_num_grass": 25,
      "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 40}, 130),   # resets + truncations inside the rollout
])
def test_fused_rollout_equals_single_steps_random_policy(over, K):
    """ppg_rollout(K) == K x ppg_step, device-side random policy with auto-reset (state, tables, observations)."""
    cfg = {**config_env, **over}
    a, b = make_env(cfg, 3, seed=21), make_env(cfg, 3, seed=21)
    a.reset()
    b.reset()
    for _ in range(K):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(K // 2, random_actions=True, auto_reset=True)
    b.rollout(K - K // 2, random_actions=True, auto_reset=True)
    _assert_same_state(_state(a), _state(b), a)


def test_fused_rollout_equals_single_steps_action_tape():
    import torch
    cfg = {**config_env, "max_steps": 25}
    K, B = 40, 2
    a, b = make_env(cfg, B, seed=5), make_env(cfg, B, seed=5)
    a.reset()
    b.reset()
    g = torch.Generator().manual_seed(0)
    tape = torch.randint(-1, 9, (K, B, a.S), generator=g, dtype=torch.int8)
    for t in range(K):
        a.step(tape[t].contiguous())
    b.rollout(K, actions=tape)
    _assert_same_state(_state(a), _state(b), a)


@pytest.mark.parametrize("coop,B", [(2, 5), (4, 3)])
def test_cooperative_fused_rollout_equals_single_steps(coop, B):
    """ppg_rollout on a handle whose plan is cooperative runs the fused form of ppgc_step (ppgc_rollout: the workgroups run on from
    step to step without a launch boundary): random policy with resets and truncations inside, and an action tape."""
    import torch
    cfg = {**config_env, "grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20, "initial_num_grass": 25,
           "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 40}
    a, b = make_env(cfg, B, seed=21), make_env(cfg, B, seed=21)
    b.set_wave_plan(4, 0, coop)
    a.reset()
    b.reset()
    for _ in range(90):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(50, random_actions=True, auto_reset=True)
    b.rollout(40, random_actions=True, auto_reset=True)
    _assert_same_state(_state(a), _state(b), a)
    cfg = {**config_env, "max_steps": 25}
    K = 35
    a, b = make_env(cfg, B, seed=5), make_env(cfg, B, seed=5)
    b.set_wave_plan(4, 0, coop)
    a.reset()
    b.reset()
    tape = torch.randint(-1, 9, (K, B, a.S), generator=torch.Generator().manual_seed(1), dtype=torch.int8)
    for t in range(K):
        a.step(tape[t].contiguous())
    b.rollout(K, actions=tape)
    _assert_same_state(_state(a), _state(b), a)


def test_emulated_random_rollout_with_drive_channels_matches_oracle():
    """drive-conditioned variant: device reset, Philox actions, auto-reset; observations incl. the drive channels
    (window sums in numpy's pairwise order) against the oracle every call."""
    cfg = {**config_env, "enable_drive_channels": True, "grid_size": 12, "initial_num_grass": 40,
           "n_initial_active_predator": 6, "n_initial_active_prey": 14, "max_steps": 60,
           "predator_hunger_safe_energy": 4.0, "energy_gain_per_step_grass": 0.2}
    env = make_env(cfg, 3)
    assert env.obs_pred.shape[2] == 7 and env.obs_prey.shape[2] == 8
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=77, n_calls=120, check_every=1)


def test_rebalance_changes_scheduling_only():
    """ppg_rebalance re-orders the assignment of envs to workgroups (heavy envs first); every result must stay the same."""
    import torch
    cfg = {**config_env, "grid_size": 12, "initial_num_grass": 40, "max_steps": 40}
    envs = [make_env(cfg, 7), make_env(cfg, 7)]
    for k, env in enumerate(envs):
        env.reset(seed=21)
        for t in range(60):
            if k == 1 and t % 5 == 0:
                env.rebalance()
            env.step(random_actions=True, auto_reset=True)
    for n in ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(envs[0], n), getattr(envs[1], n)), n
    assert torch.equal(envs[0].env_state[:, : _abi.ENV_CALLS], envs[1].env_state[:, : _abi.ENV_CALLS])
