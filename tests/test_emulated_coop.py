"""The COOPERATIVE step kernels (ppgc*_step_*: several envs per workgroup, every wavefront runs one env's transition, then all
of them write all the workgroup's observations as whole 1 KB pieces through padded cell maps) under the CPU wave emulator.
Same golden vectors and oracle as the other emulator tests; the GPU tests compare the real kernels
(test_hip_parity.py::test_cooperative_step_kernels_give_identical_results)."""
import pytest
import torch

from oracle.ppg_oracle import OracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
from tests import emu_backend
from tests.parity_utils import replay_golden_cases, rollout_vs_oracle


def maker(waves, coop, **kw):
    def make(cfg, B, **kw2):
        env = BatchedPredPreyGrass(cfg, batch_size=B, _library=emu_backend.library(), **kw, **kw2)
        env.set_wave_plan(waves, 0, coop)
        assert env.wave_plan() == (waves, 0, coop), env.wave_plan()
        return env
    return make


@pytest.mark.parametrize("waves,coop,names,max_calls", [
    (4, 4, ["default_seed0", "default_seed1"], 150),   # two envs in a workgroup with room for four: two empty env slots
    (4, 1, ["default_seed0"], 120),                     # one env, three pure helper waves
    (8, 2, ["c4_seed0"], 100),                          # 64x64 grid (eight waves: the four-map layout)
    (4, 2, ["c4_seed0"], 100),                          # 64x64 grid, four waves: THREE cell maps per env (ppgcm_step), the default there
    (4, 4, ["dense_seed0", "dense_seed3"], None),       # ghost cells / co-occupancy, many mid-step observations
    (16, 1, ["c1_seed0"], None),                        # sixteen waves on one env (small batches)
    (4, 2, ["pool_seed3"], None),                       # id pools run dry
    (4, 4, ["rewards_seed3"], None),
    (4, 4, ["seasonal_short_seed0"], 150),
    (4, 4, ["dense_rewards_seed0"], 100),               # dense reward mode reads the start-of-step energies back
])
def test_base_family_golden_cases_cooperative(waves, coop, names, max_calls):
    made = []

    def make(cfg, B, **kw):
        made.append(maker(waves, coop)(cfg, B, **kw))
        return made[-1]
    replay_golden_cases(make, names, config_env, max_calls=max_calls)
    if names == ["c4_seed0"]:
        assert made[-1].step_kernel_name() == ("ppgcm_step_q2" if waves == 4 else "ppgc8_step_q2")


@pytest.mark.parametrize("waves,coop,names,max_calls", [
    (4, 4, ["default_seed0", "default_seed1"], 150),
    (4, 4, ["dense_seed0", "dense_seed3"], None),       # ghost cells / co-occupancy, many mid-step observations
    (4, 2, ["pool_seed3"], None),
])
def test_golden_cases_cooperative_without_a_channel_0_map(waves, coop, names, max_calls, monkeypatch):
    """Round 6: the cooperative kernels have two env-region layouts -- four cell maps (every window element one lookup; what 25x25
    grids get) and THREE, channel 0 computed from the window position (what 64x64 grids get: c4_seed0 above).  PPG_COOP_MAPS=3
    forces the second one on the small grids' golden episodes."""
    monkeypatch.setenv("PPG_COOP_MAPS", "3")
    replay_golden_cases(maker(waves, coop), names, config_env, max_calls=max_calls)


def test_cooperative_launch_shape():
    lib = emu_backend.library()
    env = maker(4, 4)(dict(config_env), 6)
    env.reset(seed=1)
    env.step(random_actions=True)
    assert lib.ppg_emu_last_waves() == 4
    assert env.step_kernel_name() == "ppgc_step_q2"


@pytest.mark.parametrize("waves,coop,batch", [(4, 4, 6), (4, 3, 5), (8, 8, 9)])
def test_cooperative_random_rollout_matches_oracle_and_single_wave(waves, coop, batch):
    """Device reset + Philox actions + auto-reset (the reset runs inside the cooperative kernel too), a batch that does not fill
    the last workgroup; every call against the oracle, and the final state against the one-wave kernels."""
    cfg = {**config_env, "grid_size": 12, "n_initial_active_predator": 7, "n_initial_active_prey": 20, "initial_num_grass": 40,
           "max_steps": 40, "energy_gain_per_step_grass": 0.3}
    states = []
    for w, c in ((1, 0), (waves, coop)):
        env = maker(w, c, prey_capacity=128)(cfg, batch)
        rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=31, n_calls=60, check_grid=True)
        states.append({n: getattr(env, n).clone() for n in
                       ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey")})
        states[-1]["env_state"] = env.env_state[:, : _abi.ENV_CALLS].clone()
    for n, t in states[0].items():
        assert torch.equal(t, states[1][n]), n


def test_cooperative_big_windows_and_float32():
    """13x13 / 15x15 windows on a 9x9 grid (every window leaves the grid on all sides), float32 observations."""
    cfg = {**config_env, "grid_size": 9, "n_initial_active_predator": 5, "n_initial_active_prey": 12, "initial_num_grass": 20,
           "predator_obs_range": 13, "prey_obs_range": 15, "max_steps": 30}
    env = maker(4, 4, obs_dtype=torch.float32)(cfg, 4)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=5, n_calls=45, check_grid=True)


@pytest.mark.parametrize("name", ["rq_mixed_types_seed7", "rq_base_seed3", "rq_pool_exhaust_seed2"])
def test_second_generation_golden_cases_cooperative(name, monkeypatch):
    """The red_queen env's golden episodes (recorded PCG64 uniforms through ppg_step_uniforms) on the cooperative kernels
    (rq_mixed_types_seed7: also without a channel-0 map)."""
    from predpreygrass_amd.red_queen import BatchedRedQueen
    if name == "rq_mixed_types_seed7":
        monkeypatch.setenv("PPG_COOP_MAPS", "3")
    from tests.parity_utils_rq import replay_golden_case

    def make(cfg, B, **kw):
        env = BatchedRedQueen(cfg, batch_size=B, _library=emu_backend.library(), **kw)
        env.set_wave_plan(4, 0, 2)
        assert env.wave_plan() == (4, 0, 2) and env.step_kernel_name() == ("ppgcm2_step_q2" if name == "rq_mixed_types_seed7" else "ppgc2_step_q2")
        return env
    replay_golden_case(make, name, max_calls=120)


def test_second_generation_cooperative_rollout_matches_oracle():
    from oracle.rq_oracle import RQOracleEnv
    from predpreygrass_amd.red_queen import BatchedRedQueen
    from tests.golden_io_rq import RQGoldenCase
    from tests.parity_utils_rq import rollout_vs_oracle as rq_rollout_vs_oracle
    cfg = RQGoldenCase("rq_mixed_types_seed7").config
    env = BatchedRedQueen(cfg, batch_size=5, _library=emu_backend.library())
    env.set_wave_plan(4, 0, 4)
    n_resets, stats = rq_rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=47, n_calls=90, check_every=1, check_grid=True)
    assert stats["births"] > 3


@pytest.mark.parametrize("waves,coop", [(1, 0), (4, 0), (4, 2)])
def test_bfloat16_observation_rows_are_the_rounded_float64_rows(waves, coop):
    """obs_dtype bfloat16 (compact rows for a policy next to the env): every kernel family writes round-to-nearest-even of the
    float32 of the float64 value -- what torch's .float().bfloat16() gives -- and nothing else changes."""
    cfg = {**config_env, "max_steps": 30}
    envs = []
    for dt in (torch.float64, torch.bfloat16):
        env = maker(waves, coop, obs_dtype=dt)(cfg, 5) if waves > 1 else BatchedPredPreyGrass(cfg, batch_size=5, _library=emu_backend.library(), obs_dtype=dt)
        env.reset(seed=9)
        for _ in range(45):
            env.step(random_actions=True, auto_reset=True)
        envs.append(env)
    a, b = envs
    assert b.obs_prey.dtype == torch.bfloat16
    for n in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "grass_energy"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(a.obs_pred.float().bfloat16().view(torch.int16), b.obs_pred.view(torch.int16))
    assert torch.equal(a.obs_prey.float().bfloat16().view(torch.int16), b.obs_prey.view(torch.int16))
    assert bool((b.obs_prey.float() != 0).any())


def test_configurations_without_cooperative_kernels_fall_back():
    """Even windows, drive channels and the kickback variant keep their element-descriptor kernels: the plan drops coop_envs."""
    for extra in ({"predator_obs_range": 6}, {"enable_drive_channels": True}, {"kickback_reward_predator": 1.0}):
        env = BatchedPredPreyGrass({**config_env, **extra}, batch_size=2, _library=emu_backend.library())
        env.set_wave_plan(4, 0, 4)
        assert env.wave_plan()[2] == 0, extra


def test_allocation_choices_are_ignored_without_a_gpu():
    """obs_spread / placement candidates are allocation choices of the HIP library: on the CPU test build they change nothing."""
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    lib = emu_backend.library()
    assert not hasattr(lib, "ppg_alloc_spread") or True   # (the CPU build may or may not export it: never called here)
    a = SubBatchedPredPreyGrass(config_env, batch_size=5, n_sub=2, device="cpu", seed=3, _library=lib)
    b = SubBatchedPredPreyGrass(config_env, batch_size=5, n_sub=2, device="cpu", seed=3, _library=lib, obs_spread=8,
                                placement_candidates=3, placement_target_us=1.0)
    assert b.placement_probe_us is None and not getattr(b.subs[0], "_spread_ptrs", [])
    for g in (a, b):
        g.reset()
        for _ in range(12):
            g.step(random_actions=True, auto_reset=True)
    for x, y in zip(a.subs, b.subs):
        for n in ("row_xy", "row_energy", "row_id", "obs_pred", "obs_prey", "env_state"):
            assert torch.equal(getattr(x, n), getattr(y, n)), n
    b.subs[0].close()
    b.subs[0].close()   # idempotent


@pytest.mark.parametrize("maps", ["3", "4"])
@pytest.mark.parametrize("waves,coop", [(4, 2), (4, 4)])
def test_cooperative_spawn_fallback_on_a_crowded_grid(waves, coop, maps, monkeypatch):
    """Round 6: without a channel-0 map (PPG_COOP_MAPS=3; by default only grids too large for four maps) the spawn fallback
    (BASE:759-764: all four neighbours taken) marks occupied cells in bit 7 of the predator map's entries instead and must leave that
    map as it found it; with four maps the board is the channel-0 map.  A 6x6 grid that fills up:
    every call against the oracle (observations included), with fallback spawns actually happening."""
    monkeypatch.setenv("PPG_COOP_MAPS", maps)
    cfg = {**config_env, "grid_size": 6, "n_initial_active_predator": 6, "n_initial_active_prey": 14, "initial_num_grass": 10,
           "max_steps": 60, "energy_gain_per_step_grass": 1.5, "energy_loss_per_step_prey": 0.01, "energy_loss_per_step_predator": 0.02,
           "prey_creation_energy_threshold": 3.5, "predator_creation_energy_threshold": 6.0,
           "predator_obs_range": 5, "prey_obs_range": 7}
    env = maker(waves, coop, prey_capacity=128)(cfg, 5)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=77, n_calls=90, check_grid=True)
    es = env.env_state.cpu().numpy()
    assert es[:, _abi.ENV_FALLBACK_SPAWNS].sum() > 0 or (es[:, _abi.ENV_STATUS] & _abi.STATUS_FALLBACK_SPAWN).any()


# ---- round 6: the cooperative walls kernel (ppgc3_step: two envs per four-wave workgroup, rows written whole) -------------------
def _walls_maker(coop):
    from predpreygrass_amd.red_queen import BatchedRedQueen

    def make(cfg, B, **kw):
        env = BatchedRedQueen(cfg, batch_size=B, _library=emu_backend.library(), **{"walls": True, **kw})
        env.set_wave_plan(4, 0, coop)
        assert env.wave_plan() == (4, 0, coop), env.wave_plan()
        assert env.step_kernel_name() == "ppgc3_step_q%d" % (1 if env.prey_capacity <= 64 else 2)
        return env
    return make


@pytest.mark.parametrize("name", ["wo_zigzag_seed1", "wo_los_two_types_seed5", "wo_mask_only_shuffled_seed6"])
def test_walls_golden_cases_cooperative(name):
    """The walls env's golden episodes (walls, corner cutting, line-of-sight moves, masked observations, the visibility channel;
    recorded PCG64 uniforms) on the cooperative kernel: mid-step observations by the env's own wavefront, final rows by all four."""
    from tests.parity_utils_rq import replay_golden_case
    replay_golden_case(_walls_maker(2), name, max_calls=100)


@pytest.mark.parametrize("coop,batch", [(2, 5), (4, 6), (1, 2)])
def test_walls_cooperative_rollout_matches_oracle_with_different_walls_per_env(coop, batch):
    """Device reset around the walls + Philox actions + auto-reset; every env has its OWN walls (after the barrier a wavefront writes
    rows of envs whose bitmap is not the one of its own region), a batch that does not fill the last workgroup; every call against
    the oracle."""
    import numpy as np
    from oracle.rq_oracle import RQOracleEnv
    from tests.golden_io_rq import RQGoldenCase
    from tests.parity_utils_rq import rollout_vs_oracle as rq_rollout_vs_oracle
    case = RQGoldenCase("wo_zigzag_seed1")
    cfg = case.config
    env = _walls_maker(coop)(cfg, batch, walls=True)
    G = cfg["grid_size"]
    rng = np.random.default_rng(5)
    per_env = []
    for b in range(batch):
        extra = rng.integers(0, G, size=(6 + b, 2))
        per_env.append(np.unique(np.concatenate([np.asarray(case.wall_xy).reshape(-1, 2)[b::2], extra]), axis=0))
    env.set_walls(per_env, per_env=True)
    made = []

    def oracle():
        o = RQOracleEnv(cfg, walls=True)
        o.set_walls(per_env[len(made)])
        made.append(o)
        return o
    rq_rollout_vs_oracle(env, oracle, seed0=11, n_calls=70, check_every=1, check_grid=True)
