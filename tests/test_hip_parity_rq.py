"""GPU parity tests of the second-generation step (two agent types, move cost, energy caps, cooldown / chance /
mutation): the HIP kernels on a real MI355X, called through the C ABI, against the golden vectors of the
reference's red_queen env and against the pinned CPU oracle.  Bit-exact, no tolerance."""
import numpy as np
import pytest
import torch

from oracle.rq_oracle import RQOracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
from tests.golden_io_rq import RQGoldenCase, case_names
from tests.parity_utils_rq import replay_golden_case, rollout_vs_oracle

pytestmark = pytest.mark.gpu


def make_env(cfg, B, **kw):
    return BatchedRedQueen(cfg, batch_size=B, device="cuda:0", **kw)


@pytest.mark.parametrize("name", case_names())
def test_golden_cases_on_gpu(name):
    env, n_ordered = replay_golden_case(make_env, name)
    if "shuffled" in name:
        assert n_ordered > 10


MIXED = RQGoldenCase("rq_mixed_types_seed7").config
POOL = RQGoldenCase("rq_pool_exhaust_seed2").config
BIG_OBS = dict(config_env_base, predator_obs_range=9, prey_obs_range=11, n_initial_active_type_2_predator=4,
               n_possible_type_2_predators=200, mutation_rate_predator=0.2, mutation_rate_prey=0.2)


@pytest.mark.parametrize("cfg,B,calls,cap,every", [
    (config_env_base, 64, 300, 128, 20),   # the reference's base config of this env
    (MIXED, 64, 300, 128, 5),              # both types of both species, typed rewards, caps, cooldown 3
    (POOL, 32, 250, 64, 5),                # id pools run dry: reward without a child
    (BIG_OBS, 16, 120, 256, 10),           # generic observation geometry (descriptors in LDS), 4 prey registers
])
def test_random_rollout_matches_oracle_on_gpu(cfg, B, calls, cap, every):
    env = make_env(cfg, B, prey_capacity=cap)
    n_resets, stats = rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=4242, n_calls=calls, check_every=every,
                                        check_grid=True)
    assert stats["births"] > 0


def make_coop(coop):
    def make(cfg, B, **kw):
        env = make_env(cfg, B, **kw)
        env.set_wave_plan(4, 0, coop)
        assert env.wave_plan() == (4, 0, coop) and env.step_kernel_name().startswith("ppgc2_step_q")
        return env
    return make


@pytest.mark.parametrize("name", [n for n in case_names() if "shuffled" not in n])
def test_golden_cases_through_the_cooperative_kernels(name):
    """The red_queen env's golden episodes (recorded uniforms, ppg_step_uniforms) on ppgc2_step: two envs per four-wave workgroup."""
    replay_golden_case(make_coop(2), name)


@pytest.mark.parametrize("cfg,B,calls,cap,coop", [(config_env_base, 63, 300, 128, 2), (MIXED, 64, 300, 128, 4), (POOL, 31, 250, 64, 3)])
def test_cooperative_random_rollout_matches_oracle_on_gpu(cfg, B, calls, cap, coop):
    env = make_coop(coop)(cfg, B, prey_capacity=cap)
    n_resets, stats = rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=4243, n_calls=calls, check_every=5, check_grid=True)
    assert stats["births"] > 0


@pytest.mark.parametrize("cfg,B,coop,cap", [(config_env_base, 4096, 2, 128), (MIXED, 130, 2, 128), (POOL, 61, 4, 64)])
def test_fused_rollout_equals_single_steps_on_gpu(cfg, B, coop, cap):
    """ppg_rollout on a second-generation handle (ppgc2_rollout: the fused form of the cooperative kernel) against the same number of
    ppg_step launches: random policy, reproduction from the device's Philox uniforms, truncations and auto-resets inside the
    launches; at 4096 envs the workgroups drift apart by whole steps.  Then an action tape."""
    cfg = dict(cfg, max_steps=60)
    a, b = make_env(cfg, B, prey_capacity=cap, seed=77), make_env(cfg, B, prey_capacity=cap, seed=77)
    b.set_wave_plan(4, 0, coop)
    a.reset()
    b.reset()
    for _ in range(150):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(90, random_actions=True, auto_reset=True)
    b.rollout(60, random_actions=True, auto_reset=True)
    names = ("row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "row_lastrep", "grass_energy",
             "obs_pred", "obs_prey")
    for n in names:
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(a.env_state[:, : _abi.ENV_CALLS], b.env_state[:, : _abi.ENV_CALLS])
    K = 40
    tape = torch.randint(-1, 9, (K, B, a.S), generator=torch.Generator().manual_seed(4), dtype=torch.int8).to("cuda:0")
    for t in range(K):
        a.step(tape[t])
    b.rollout(K, actions=tape)
    for n in names:
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    env = make_env(cfg, 4)
    env.set_wave_plan(1, 0, 0)
    env.reset()
    with pytest.raises(RuntimeError, match="cooperative"):
        env.rollout(2, random_actions=True)


def test_float64_observations_on_gpu():
    env = make_env(MIXED, 8, obs_dtype=torch.float64)
    rollout_vs_oracle(env, lambda: RQOracleEnv(MIXED), seed0=5, n_calls=80, check_every=4)


def test_full_size_4096_envs_sampled_oracle_and_determinism():
    cfg = dict(config_env_base)
    B = 4096
    env = make_env(cfg, B)
    sample = list(range(0, B, 256))
    rollout_vs_oracle(env, lambda: RQOracleEnv(cfg), seed0=99, n_calls=300, check_every=60, envs=sample)
    es = env.env_state.cpu().numpy()
    assert (es[:, _abi.ENV_STATUS] & ~_abi.STATUS_FALLBACK_SPAWN == 0).all()
    env2 = make_env(cfg, B)
    env2.set_seeds(99)
    env2.env_state.zero_()
    env2.env_state[:, _abi.ENV_FLAGS] = _abi.ENVF_DONE
    env2.env_state[:, _abi.ENV_EPISODE] = -1
    for _ in range(300):
        env2.step(random_actions=True, auto_reset=True)
    for name in ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_lastrep", "env_state", "obs_pred", "obs_prey"):
        a, b = getattr(env, name), getattr(env2, name)
        if name == "env_state":
            a, b = a[:, : _abi.ENV_CALLS], b[:, : _abi.ENV_CALLS]
        assert torch.equal(a, b), name


@pytest.mark.parametrize("seed", range(100, 130))
def test_random_gen2_config_matches_oracle_on_gpu(seed):
    from predpreygrass_amd.red_queen import PredPreyGrass
    from tests.test_rq_random_configs import run_differential
    run_differential(lambda cfg: PredPreyGrass(cfg, device="cuda:0"), seed)


# ---- walls_occlusion variant (ppg3_* kernels) -------------------------------------------------------------------

@pytest.mark.parametrize("name", case_names(walls=True))
def test_walls_golden_cases_on_gpu(name):
    replay_golden_case(make_env, name)


@pytest.mark.parametrize("waves", [None, 1, 2, 4])
def test_walls_random_rollout_matches_oracle_on_gpu(waves):
    """ppg3_step, ppgwp3_step (two wavefronts per env) and ppgw3_step (four); None = what the library picks."""
    case = RQGoldenCase("wo_los_two_types_seed5")
    cfg, walls = case.config, case.wall_xy

    def oracle():
        o = RQOracleEnv(cfg, walls=True)
        o.set_walls(walls)
        return o
    env = make_env(cfg, 64, walls=True)
    if waves is not None:
        env.set_wave_plan(waves)
        assert env.wave_plan()[0] == waves
    env.set_walls(walls)
    n_resets, stats = rollout_vs_oracle(env, oracle, seed0=31, n_calls=250, check_every=5, check_grid=True)
    assert stats["births"] > 20 and n_resets > 10


def make_walls_coop(coop):
    def make(cfg, B, **kw):
        env = make_env(cfg, B, **{"walls": True, **kw})
        env.set_wave_plan(4, 0, coop)
        assert env.wave_plan() == (4, 0, coop) and env.step_kernel_name().startswith("ppgc3_step_q"), (env.wave_plan(), env.step_kernel_name())
        return env
    return make


@pytest.mark.parametrize("name", case_names(walls=True))
def test_walls_golden_cases_through_the_cooperative_kernel(name):
    """Round 6: the walls env's golden episodes on ppgc3_step -- two envs per four-wave workgroup, mid-step observations by the env's
    own wavefront, the final rows by all four (even windows included: wo_mask_only_shuffled_seed6)."""
    replay_golden_case(make_walls_coop(2), name)


@pytest.mark.parametrize("coop,B", [(2, 63), (4, 64), (3, 31)])
def test_walls_cooperative_random_rollout_with_walls_of_its_own_per_env(coop, B):
    """Every env with its OWN walls: after the workgroup's barrier a wavefront writes rows of envs whose bitmap, line-of-sight masks and
    staging areas are not those of the env it stepped.  Every fifth call against the oracle (tables, observations, grid)."""
    case = RQGoldenCase("wo_zigzag_seed1")
    cfg, G = case.config, case.config["grid_size"]
    rng = np.random.default_rng(17)
    base = np.asarray(case.wall_xy).reshape(-1, 2)
    per_env = [np.unique(np.concatenate([base[b % 3::3], rng.integers(0, G, size=(5 + b % 11, 2))]), axis=0) for b in range(B)]
    env = make_walls_coop(coop)(cfg, B)
    env.set_walls(per_env, per_env=True)
    made = []

    def oracle():
        o = RQOracleEnv(cfg, walls=True)
        o.set_walls(per_env[len(made)])
        made.append(o)
        return o
    n_resets, stats = rollout_vs_oracle(env, oracle, seed0=77, n_calls=200, check_every=5, check_grid=True)
    assert stats["births"] > 20


@pytest.mark.parametrize("coop", [0, 2])
def test_shared_and_per_env_visibility_masks_across_wall_changes_on_gpu(coop):
    """ppg_walls_changed: one wall layout for the batch = every env reads env 0's masks; one env's walls changed = per-env tables again
    (and back): at every stage the same bits as a handle that walks the lines itself (no precomputed masks)."""
    case = RQGoldenCase("wo_los_two_types_seed5")
    B = 130
    half = case.wall_xy[: len(case.wall_xy) // 2]
    layouts = ([case.wall_xy] * B, [case.wall_xy if b != 77 else half for b in range(B)], [case.wall_xy[:7]] * B)
    envs = []
    for pre in (True, False):
        e = make_walls_coop(coop)(case.config, B, seed=11) if coop else make_env(case.config, B, walls=True, seed=11)
        e.set_walls(layouts[0], per_env=True, precompute_visibility=pre)
        e.reset()
        envs.append(e)
    for stage, layout in enumerate(layouts):
        if stage:
            for e, pre in zip(envs, (True, False)):
                e.set_walls(layout, per_env=True, precompute_visibility=pre)
                e.reset()
        for _ in range(25):
            for e in envs:
                e.step(random_actions=True, auto_reset=True)
            for n in ("obs_pred", "obs_prey", "row_xy", "row_energy", "row_info", "env_state"):
                assert torch.equal(getattr(envs[0], n), getattr(envs[1], n)), (stage, n)


def test_walls_cooperative_kernel_gives_what_the_four_wave_kernel_gives_at_full_size():
    """4096 envs of the reference's zigzag layout, every line-of-sight option on: 150 calls on ppgc3_step and on ppgw3_step, every table
    and every observation row in use bit-identical."""
    from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls as cfg
    states = []
    for coop in (0, 2):
        env = make_env(cfg, 4096, walls=True)
        env.set_wave_plan(4, 0, coop)
        assert env.step_kernel_name() == ("ppgc3_step_q2" if coop else "ppgw3_step_q2")
        env.set_walls(cfg["manual_wall_positions"])
        env.set_seeds(99)
        env.env_state.zero_()
        env.env_state[:, _abi.ENV_FLAGS] = _abi.ENVF_DONE   # (the first call is the device reset)
        env.env_state[:, _abi.ENV_EPISODE] = -1
        for _ in range(150):
            env.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        states.append(env)
    a, b = states
    assert torch.equal(a.env_state[:, : _abi.ENV_CALLS], b.env_state[:, : _abi.ENV_CALLS])
    nP, nQ = a.env_state[:, _abi.ENV_N_PRED_ROWS], a.env_state[:, _abi.ENV_N_PREY_ROWS]
    cp = a.pred_capacity
    rows = torch.arange(a.S, device="cuda:0")[None, :]
    used = (rows < nP[:, None]) | ((rows >= cp) & (rows < cp + nQ[:, None]))
    for n in ("row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "row_info"):
        assert torch.equal(getattr(a, n)[used], getattr(b, n)[used]), n
    assert torch.equal(a.obs_pred[used[:, :cp]], b.obs_pred[used[:, :cp]])
    assert torch.equal(a.obs_prey[used[:, cp:]], b.obs_prey[used[:, cp:]])
    assert int(used.sum()) > 4096 * 10


@pytest.mark.parametrize("seed", range(200, 220))
def test_random_walls_config_matches_oracle_on_gpu(seed):
    from predpreygrass_amd.walls_occlusion import PredPreyGrass as WallsEnv
    from tests.test_rq_random_configs import run_differential
    run_differential(lambda cfg: WallsEnv(cfg, device="cuda:0"), seed, walls=True)


# ---- the per-agent analytics of the dict classes (RQ:99-116) ------------------------------------------------------

def _make_dict_env_on_gpu(case):
    if case.walls:
        from predpreygrass_amd.walls_occlusion import PredPreyGrass as Env
    else:
        from predpreygrass_amd.red_queen import PredPreyGrass as Env
    return Env(case.config, device="cuda:0", _check_analytics=True)


@pytest.mark.parametrize("name", ["rq_mixed_types_seed7", "rq_pool_exhaust_seed2", "rq_shuffled_seed5", "rq_base_seed3",
                                  "wo_los_two_types_seed5", "wo_mask_only_shuffled_seed6"])
def test_analytics_books_equal_the_references_on_gpu(name):
    """unique_agent_stats / death_agents_stats / per_step_agent_data / ages / offspring lists after whole golden episodes stepped by
    the HIP kernels == the books of the reference itself (tests/golden/*/<name>.analytics.json.gz), floats bit for bit."""
    from tests.test_rq_analytics import replay_analytics
    replay_analytics(_make_dict_env_on_gpu, name)
