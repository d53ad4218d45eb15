"""GPU parity tests: the HIP path (libppg_hip.so on a real MI355X, called through the C ABI) against
the golden vectors of the reference and against the CPU oracle.  Bit-exact everywhere
(integer/index work and IEEE float64 sums; no tolerance)."""
import numpy as np
import pytest
import torch

from oracle.ppg_oracle import OracleEnv
from predpreygrass_amd import _abi
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
from tests.parity_utils import replay_golden_cases, rollout_vs_oracle

pytestmark = pytest.mark.gpu


def make_env(cfg, B, **kw):
    return BatchedPredPreyGrass(cfg, batch_size=B, device="cuda:0", **kw)


@pytest.mark.parametrize("names", [
    ["c1_seed0"], ["default_seed0", "default_seed1"], ["c4_seed0"], ["dense_seed0", "dense_seed3"],
    ["rewards_seed3"], ["pool_seed3"], ["even_obs_seed0"],
    ["seasonal_short_seed0"], ["seasonal_default_seed1"], ["plus_eating_seed2"],
    ["dense_rewards_seed0"], ["dense_additive_seed4"], ["kickback_seed0"], ["kickback_fast_seed5"],
    ["drive_default_seed2"], ["drive_custom_lists_big_windows_seed4"],
])
def test_golden_cases_on_gpu(names):
    replay_golden_cases(make_env, names, config_env)


C1 = {"n_initial_active_predator": 4, "n_initial_active_prey": 8, "initial_num_grass": 30}
C4 = {"grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32, "predator_obs_range": 7,
      "prey_obs_range": 7}
DENSE = {"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20, "initial_num_grass": 25,
         "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 150}
TINY = {"grid_size": 5, "n_initial_active_predator": 5, "n_initial_active_prey": 9, "initial_num_grass": 8,
        "predator_obs_range": 3, "prey_obs_range": 5, "max_steps": 60, "energy_gain_per_step_grass": 0.5}


@pytest.mark.parametrize("over,B,calls,cap,every", [
    (C1, 1, 150, 64, 1),            # BASELINE config 1
    ({}, 256, 300, 128, 25),        # BASELINE config 2: 256 envs, default config
    (C4, 32, 120, 128, 10),         # BASELINE config 4 geometry (64x64, obs 7x7)
    (DENSE, 64, 250, 128, 5),       # co-occupancy / ghost cells / spawn fallback
    (TINY, 64, 250, 256, 5),        # 5x5 grid: fallback spawns, extinction, resets every few steps
])
def test_random_rollout_matches_oracle_on_gpu(over, B, calls, cap, every):
    cfg = {**config_env, **over}
    env = make_env(cfg, B, prey_capacity=cap)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=777, n_calls=calls, check_every=every, check_grid=True)


def test_float32_observations_on_gpu():
    cfg = dict(config_env)
    env = make_env(cfg, 8, obs_dtype=torch.float32)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=5, n_calls=60, check_every=5)


def test_full_size_4096_envs_properties_and_sampled_oracle():
    """BASELINE config 3 (4096 envs x 25x25): sampled envs are compared with the oracle on every 50th
    call; for all envs size-independent invariants are checked."""
    cfg = dict(config_env)
    B = 4096
    env = make_env(cfg, B)
    sample = list(range(0, B, 128))
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=31337, n_calls=400, check_every=50, envs=sample)
    es = env.env_state.cpu().numpy()
    assert (es[:, _abi.ENV_STATUS] & ~_abi.STATUS_FALLBACK_SPAWN == 0).all()
    # two runs with the same seeds are identical (determinism under massive parallelism)
    env2 = make_env(cfg, B)
    env2.set_seeds(31337)
    env2.env_state.zero_()
    env2.env_state[:, _abi.ENV_FLAGS] = _abi.ENVF_DONE
    env2.env_state[:, _abi.ENV_EPISODE] = -1
    for _ in range(400):
        env2.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.equal(env.env_state[:, : _abi.ENV_CALLS], env2.env_state[:, : _abi.ENV_CALLS])
    nP, nQ = env.env_state[:, _abi.ENV_N_PRED_ROWS], env.env_state[:, _abi.ENV_N_PREY_ROWS]
    cp = env.pred_capacity
    mp = torch.arange(cp, device="cuda:0")[None, :] < nP[:, None]                    # predator rows in use
    mq = torch.arange(env.prey_capacity, device="cuda:0")[None, :] < nQ[:, None]     # prey rows in use
    for name in ("row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward"):
        a, b = getattr(env, name), getattr(env2, name)
        assert torch.equal(a[:, :cp][mp], b[:, :cp][mp]) and torch.equal(a[:, cp:][mq], b[:, cp:][mq]), name
    for name in ("grass_xy", "grass_energy"):
        assert torch.equal(getattr(env, name), getattr(env2, name)), name
    assert torch.equal(env.obs_pred[mp], env2.obs_pred[mp]) and torch.equal(env.obs_prey[mq], env2.obs_prey[mq])
    # invariants: alive counts match flags; every live agent inside the grid; energies of live agents > 0
    G = cfg["grid_size"]
    flags = env.row_flags.cpu().numpy()
    xy = env.row_xy.cpu().numpy().astype(np.int64)
    en = env.row_energy.cpu().numpy()
    for b in range(0, B, 37):
        nP, nQ = es[b, _abi.ENV_N_PRED_ROWS], es[b, _abi.ENV_N_PREY_ROWS]
        rows = list(range(nP)) + list(range(env.pred_capacity, env.pred_capacity + nQ))
        alive = [r for r in rows if not flags[b, r] & _abi.ROW_DIED]
        assert len([r for r in alive if r < env.pred_capacity]) == es[b, _abi.ENV_N_PRED_ALIVE]
        assert len([r for r in alive if r >= env.pred_capacity]) == es[b, _abi.ENV_N_PREY_ALIVE]
        for r in alive:
            assert 0 <= (xy[b, r] >> 8) < G and 0 <= (xy[b, r] & 255) < G
            assert en[b, r] > 0
    # obs channel 0 is a 0/1 mask and the observer's own energy sits at the window centre
    b = 5
    recs = env.records(b)
    op, oq = env.obs_pred[b].cpu().numpy(), env.obs_prey[b].cpu().numpy()
    for name, ty, row, _, te, _ in recs:
        o = (op if ty == 0 else oq)[row]
        assert set(np.unique(o[0]).tolist()) <= {0.0, 1.0}
        if not te:
            c = (o.shape[1] - 1) // 2
            s = row if ty == 0 else env.pred_capacity + row
            if flags[b, s] & _abi.ROW_OWNS:
                assert o[1 + ty, c, c] == en[b, s]


def test_dict_api_on_gpu_including_shuffled_action_order():
    """The reference-shaped class on the real device; dense_shuffled_seed17 drives the explicit-order
    kernel variant (ppg_step_ordered)."""
    from predpreygrass_amd.env import PredPreyGrass
    from tests.test_env_api import replay_through_dict_api
    mk = lambda cfg: PredPreyGrass(cfg, device="cuda:0")
    replay_through_dict_api("c1_seed0", mk)
    replay_through_dict_api("dense_shuffled_seed17", mk)
    replay_through_dict_api("default_seed0", mk, max_calls=120)


def test_sub_batches_on_streams_equal_one_batch():
    """4096 envs as 3 sub-batches on 3 streams == one batch of 4096 (same seeds), state and observations."""
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    cfg = dict(config_env)
    B = 4096
    one = make_env(cfg, B, seed=11)
    one.reset()
    grp = SubBatchedPredPreyGrass(cfg, batch_size=B, n_sub=3, device="cuda:0", seed=11)
    grp.reset()
    for _ in range(150):
        one.step(random_actions=True, auto_reset=True)
        grp.step(random_actions=True, auto_reset=True)
    grp.synchronize()
    torch.cuda.synchronize()
    for name in ("env_state", "row_xy", "row_energy", "row_flags", "grass_energy"):
        cat = torch.cat([getattr(e, name) for e in grp.subs])
        ref = getattr(one, name)
        if name == "env_state":
            assert torch.equal(cat[:, :13], ref[:, :13])
        elif name.startswith("row"):
            n = one.env_state[:, _abi.ENV_N_PRED_ROWS]
            mask = torch.arange(one.S, device="cuda:0")[None, :] < n[:, None]   # predator rows in use
            assert torch.equal(cat[mask], ref[mask]), name
        else:
            assert torch.equal(cat, ref), name
    nq = one.env_state[:, _abi.ENV_N_PREY_ROWS]
    mq = torch.arange(one.prey_capacity, device="cuda:0")[None, :] < nq[:, None]
    assert torch.equal(torch.cat([e.obs_prey for e in grp.subs])[mq], one.obs_prey[mq])


def test_full_size_64x64_config4_sampled_oracle():
    """BASELINE config 4: 4096 envs x 64x64, 16 predators / 32 prey, obs 7x7 (LDS-tile stress)."""
    cfg = {**config_env, **C4}
    env = make_env(cfg, 4096)
    assert env.lds_bytes < 64 * 1024
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=4242, n_calls=90, check_every=30, envs=list(range(0, 4096, 512)))
    es = env.env_state.cpu().numpy()
    assert (es[:, _abi.ENV_STATUS] & ~_abi.STATUS_FALLBACK_SPAWN == 0).all()


@pytest.mark.parametrize("seed", range(1000, 1030))
def test_random_config_matches_oracle_on_gpu(seed):
    """Random configurations x partial, shuffled action dicts (see tests/test_random_configs.py), on the device."""
    from predpreygrass_amd.env import PredPreyGrass
    from tests.test_random_configs import run_differential
    run_differential(lambda cfg: PredPreyGrass(cfg, device="cuda:0"), seed)


def test_fused_rollout_equals_single_steps_on_gpu():
    """ppg_rollout(K) == K x ppg_step on the device, 512 envs, resets inside the rollout."""
    from tests.test_emulated_kernel import _assert_same_state, _state
    cfg = {**config_env, "max_steps": 60}
    a, b = make_env(cfg, 512, seed=77), make_env(cfg, 512, seed=77)
    a.reset()
    b.reset()
    for _ in range(150):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(100, random_actions=True, auto_reset=True)
    b.rollout(50, random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    sa = {k: v.cpu() for k, v in _state(a).items()}
    sb = {k: v.cpu() for k, v in _state(b).items()}
    _assert_same_state(sa, sb, a)


@pytest.mark.parametrize("idx", range(3))
def test_maximum_size_configs_on_gpu(idx):
    from predpreygrass_amd.env import PredPreyGrass
    from tests.test_random_configs import BIG_CONFIGS, run_big
    run_big(lambda cfg: PredPreyGrass(cfg, device="cuda:0", prey_capacity=256 if idx == 0 else None),
            BIG_CONFIGS[idx], seed=idx)


def test_32768_envs_on_one_gpu_sampled_oracle():
    """BASELINE config 5's 32768 envs on ONE device (8x the per-GPU shard): strides and indices beyond 32 bits
    of bytes (obs_prey alone is 10.9 GB), sampled envs vs oracle, status clean everywhere."""
    cfg = dict(config_env)
    B = 32768
    env = make_env(cfg, B)
    sample = [0, 1, 4095, 4096, 16383, 16384, 32766, 32767]
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=2 ** 40 + 7, n_calls=120, check_every=40, envs=sample)
    es = env.env_state.cpu().numpy()
    assert (es[:, _abi.ENV_STATUS] & ~_abi.STATUS_FALLBACK_SPAWN == 0).all()
    assert (es[:, _abi.ENV_CALLS] == 120).all()
    del env
    torch.cuda.empty_cache()


def test_drive_conditioned_variant_on_gpu():
    """drive channels (window sums in numpy's pairwise order): random rollouts vs the oracle, 64 envs, and the shuffled
    golden episode through the drive-conditioned dict class."""
    cfg = {**config_env, "enable_drive_channels": True}
    env = make_env(cfg, 64)
    assert env.obs_pred.shape[2] == 7 and env.obs_prey.shape[2] == 8
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=99, n_calls=200, check_every=10)
    from predpreygrass_amd.drive_conditioned import PredPreyGrass as DriveEnv
    from tests.test_env_api import replay_through_dict_api
    replay_through_dict_api("drive_dense_seed3", lambda c, **kw: DriveEnv({k: v for k, v in c.items() if k != "enable_drive_channels"},
                                                                       device="cuda:0", **kw))


@pytest.mark.parametrize("cap,G,prey0,grass0", [(64, 14, 8, 30), (128, 14, 80, 100), (256, 18, 170, 140)])
@pytest.mark.parametrize("multi", [1, 2, 4])
def test_drive_variant_every_register_count_on_gpu(cap, G, prey0, grass0, multi):
    """ppg4_step_q{1,2,4}, ppgwp4_step_q{1,2,4} and ppgw4_step_q{1,2,4}: 64 / 128 / 256 prey rows per env (1, 2, 4 prey registers;
    the configs start with 8 / 80 / 170 prey so that the upper registers are in use), one, two and four waves per env; every fourth
    call against the oracle."""
    cfg = {**config_env, "enable_drive_channels": True, "grid_size": G, "initial_num_grass": grass0,
           "n_initial_active_predator": 6, "n_initial_active_prey": prey0, "max_steps": 80,
           "predator_creation_energy_threshold": 30.0, "prey_creation_energy_threshold": 12.0}
    if cap == 64:
        cfg.update(energy_gain_per_step_grass=0.4, predator_creation_energy_threshold=12.0, prey_creation_energy_threshold=8.0)
    env = make_env(cfg, 16, prey_capacity=cap)
    env.set_wave_plan(multi)
    assert env.wave_plan()[0] == multi
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=1234 + cap, n_calls=100, check_every=4)
    assert not (env.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_PREY_OVERFLOW).any()


@pytest.mark.parametrize("cls_name", ["base", "red_queen"])
def test_multiwave_step_kernels_give_identical_results(cls_name):
    """One, four and eight wavefronts per env (ppg_step / ppgw_step / ppgw8_step; the library picks by batch size,
    ppg_set_wave_plan forces a choice) must produce the same tables and observations, bit for bit."""
    if cls_name == "base":
        mk = lambda: make_env(dict(config_env), 300)
    else:
        from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
        mk = lambda: BatchedRedQueen(config_env_base, batch_size=300, device="cuda:0")
    results = []
    for waves in (1, 4, 8):
        env = mk()
        env.set_wave_plan(waves)
        assert env.wave_plan()[0] == waves
        env.reset(seed=11)
        for _ in range(150):
            env.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        results.append({n: getattr(env, n).clone() for n in
                        ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "env_state",
                         "grass_energy", "obs_pred", "obs_prey")})
    for other in results[1:]:
        for n, t in results[0].items():
            a, b = (t[:, : _abi.ENV_CALLS], other[n][:, : _abi.ENV_CALLS]) if n == "env_state" else (t, other[n])
            assert torch.equal(a, b), n


def make_coop(waves, coop, **kw):
    def make(cfg, B, **kw2):
        env = make_env(cfg, B, **kw, **kw2)
        env.set_wave_plan(waves, 0, coop)
        assert env.wave_plan() == (waves, 0, coop)
        return env
    return make


@pytest.mark.parametrize("waves,coop,names", [
    (4, 4, ["default_seed0", "default_seed1"]), (4, 4, ["dense_seed0", "dense_seed3"]), (8, 2, ["c4_seed0"]), (16, 1, ["c1_seed0"]),
    (4, 3, ["pool_seed3"]), (8, 8, ["rewards_seed3"]), (4, 4, ["seasonal_default_seed1"]), (4, 4, ["dense_additive_seed4"]),
])
def test_golden_cases_through_the_cooperative_kernels(waves, coop, names):
    """The reference's golden episodes through ppgc*_step (several envs per workgroup, padded maps, 1 KB observation pieces)."""
    replay_golden_cases(make_coop(waves, coop), names, config_env)


@pytest.mark.parametrize("over,B,calls,cap,waves,coop", [
    ({}, 256, 300, 128, 4, 4), (C4, 30, 120, 128, 8, 2), (DENSE, 61, 250, 128, 4, 4), (TINY, 64, 250, 128, 8, 8), ({}, 37, 200, 64, 16, 1),
])
def test_cooperative_random_rollout_matches_oracle_on_gpu(over, B, calls, cap, waves, coop):
    cfg = {**config_env, **over}
    env = make_coop(waves, coop, prey_capacity=cap)(cfg, B)
    rollout_vs_oracle(env, lambda: OracleEnv(cfg), seed0=778, n_calls=calls, check_every=5, check_grid=True)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.bfloat16])
def test_cooperative_step_kernels_give_identical_results(dtype):
    """ppgc_step (4 envs x 4 waves), ppgc8_step, ppgc16_step and odd env counts per workgroup against the one-wave kernel on 1000
    envs x 200 calls with auto-reset: tables and observations bit for bit.  bfloat16 rows: the four-wave kernel is its 64-register
    build ppgch_step (eight workgroups per CU)."""
    names = ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "env_state", "grass_energy", "obs_pred", "obs_prey")
    results = []
    for waves, coop in ((1, 0), (4, 4), (8, 8), (16, 5), (4, 1), (4, 2)):
        env = make_env(dict(config_env), 1000, obs_dtype=dtype)
        env.set_wave_plan(waves, 0, coop)
        assert env.wave_plan() == (waves, 0, coop)
        if waves == 4:   # (eight workgroups of one or two envs fit a CU's LDS, of four they do not)
            assert env.step_kernel_name() == ("ppgch_step_q2" if dtype == torch.bfloat16 and coop <= 2 else "ppgc_step_q2")
        env.reset(seed=12)
        for _ in range(200):
            env.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        results.append({n: getattr(env, n).clone() for n in names})
        del env
    for other in results[1:]:
        for n, t in results[0].items():
            a, b = (t[:, : _abi.ENV_CALLS], other[n][:, : _abi.ENV_CALLS]) if n == "env_state" else (t, other[n])
            assert torch.equal(a, b), n


def test_cooperative_fused_rollout_equals_single_steps_on_gpu():
    """ppg_rollout on a cooperative plan (ppgc_rollout: the workgroups run on from step to step, no launch boundary) against the
    same number of ppg_step calls: 4096 envs x 150 steps with resets inside, and an action tape on 300 envs."""
    cfg = dict(config_env)
    a, b = make_env(cfg, 4096, seed=77), make_env(cfg, 4096, seed=77)
    assert b.wave_plan() == (4, 0, 2)
    a.set_wave_plan(1)
    a.reset()
    b.reset()
    for _ in range(150):
        a.step(random_actions=True, auto_reset=True)
    b.rollout(100, random_actions=True, auto_reset=True)
    b.rollout(50, random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    for n in ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(a.env_state[:, : _abi.ENV_CALLS], b.env_state[:, : _abi.ENV_CALLS])
    del a, b
    cfg = {**config_env, "max_steps": 30}
    K, B = 45, 300
    a, b = make_env(cfg, B, seed=5), make_env(cfg, B, seed=5)
    b.set_wave_plan(4, 0, 4)
    a.reset()
    b.reset()
    tape = torch.randint(-1, 9, (K, B, a.S), generator=torch.Generator().manual_seed(3), dtype=torch.int8).cuda()
    for t in range(K):
        a.step(tape[t].contiguous())
    b.rollout(K, actions=tape)
    torch.cuda.synchronize()
    for n in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n


def test_default_wave_plans_give_identical_results():
    """What the library picks by itself -- sixteen waves per env at 200 envs, the cooperative kernel (two envs per four-wave
    workgroup) for a full GPU of 25x25 and of 64x64 grids, a pair of waves with 8-bit maps for a full GPU of 80x80 grids, the second
    generation's cooperative kernel -- against the one-wave kernels, bit for bit."""
    from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
    c4 = {**config_env, **C4}
    cases = [(lambda: make_env(dict(config_env), 200), b"ppgw16_step_q2", 120),
             (lambda: make_env(dict(config_env), 4096), b"ppgc_step_q2", 60),
             (lambda: make_env(c4, 4096), b"ppgcm_step_q2", 40),       # (round 6: THREE cell maps per env -> four cooperative workgroups per CU)
             (lambda: make_env({**c4, "grid_size": 80}, 4096), b"ppgwp_step_q2", 30),
             (lambda: BatchedRedQueen(config_env_base, batch_size=4096, device="cuda:0"), b"ppgc2_step_q2", 120)]
    names = ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "env_state", "grass_energy", "obs_pred", "obs_prey")
    for mk, kernel, calls in cases:
        results = []
        for force in (None, 1):
            env = mk()
            if force is None:
                assert env._lib.ppg_step_kernel_name(env._handle) == kernel
            else:
                env.set_wave_plan(force)
                assert env.wave_plan() == (1, 0, 0)
            env.reset(seed=21)
            for _ in range(calls):
                env.step(random_actions=True, auto_reset=True)
            torch.cuda.synchronize()
            results.append({n: getattr(env, n).clone() for n in names})
            del env
        for n in names:
            a, b = results[0][n], results[1][n]
            if n == "env_state":
                a, b = a[:, : _abi.ENV_CALLS], b[:, : _abi.ENV_CALLS]
            assert torch.equal(a, b), (kernel, n)
        del results
        torch.cuda.empty_cache()


def test_bench_rccl_gather_legs_with_one_rank():
    """bench.py under torch.distributed.run with a world of ONE rank and --force-dist: process-group init on RCCL,
    barrier / max-over-ranks timing, the synchronous and the overlapped single-collective observation-gather legs on the
    real device (ppg_pack + one all_gather_into_tensor per step)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(port, timeout):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "50",
               "--envs", "1024", "--force-dist", "--no-cpu-baseline", "--gather-steps", "5", "--preroll-max", "400"]
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)
    try:
        out = run(29533, 420)   # (takes ~10 s)
    except subprocess.TimeoutExpired as first:
        # Seen ONCE in round 6, on a lease whose every figure was off (the pool's slow state): the launcher did not come back.  One
        # retry on another port; a second time-out fails the test with what the first attempt had printed.
        try:
            out = run(29541, 420)
        except subprocess.TimeoutExpired as second:
            tail = lambda e: ((e.stderr or b"")[-1500:].decode(errors="replace") if isinstance(e.stderr, bytes) else str(e.stderr or "")[-1500:])
            pytest.fail("torchrun + bench.py with one rank timed out twice; stderr tails:\n" + tail(first) + "\n----\n" + tail(second))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e6
    assert d["roofline"]["achieved"] <= d["roofline"]["peak"]
    for leg in ("obs_gather", "obs_gather_overlapped"):
        assert "error" not in d[leg], d[leg]
        assert d[leg]["collectives_per_step"] == 1 and d[leg]["image_overflows"] == 0
        assert d[leg]["wire_bytes_per_step_per_rank"] > 1e6
    # the other modes on RCCL too: gather-to-root, the all-pairs spelling, the image without observations
    for leg in ("obs_gather_to_root", "obs_all_pairs", "ids_rewards_gather"):
        assert "error" not in d[leg], d[leg]
        assert d[leg]["collectives_per_step"] == 1 and d[leg]["image_overflows"] == 0
    assert d["ids_rewards_gather"]["wire_bytes_per_step_per_rank"] * 20 < d["obs_gather"]["wire_bytes_per_step_per_rank"]


def test_snapshot_restore_on_gpu_base_and_second_generation():
    """get_state_snapshot / restore_state_snapshot through ppg_export_state / ppg_import_state on the device
    (BASE:768-804, evaluate_ppo_from_checkpoint_debug.py:164-182): snapshot at call 20 -> 10 steps -> restore -> the same 10
    steps bit-identical AND equal to the reference's golden episode; red_queen incl. its PCG64 stream; walls variant."""
    from predpreygrass_amd.env import PredPreyGrass
    from tests.test_env_api import check_snapshot_against_golden, check_state_image_moves_between_handles
    check_snapshot_against_golden(lambda cfg, **kw: PredPreyGrass(cfg, device="cuda:0", **kw))
    check_state_image_moves_between_handles(lambda cfg, **kw: PredPreyGrass(cfg, device="cuda:0", **kw))
    from predpreygrass_amd import red_queen, walls_occlusion
    from tests.test_rq_env_api import check_rq_snapshot_against_golden
    check_rq_snapshot_against_golden(lambda cfg, **kw: red_queen.PredPreyGrass(cfg, device="cuda:0", **kw), "rq_mixed_types_seed7")
    check_rq_snapshot_against_golden(lambda cfg, **kw: walls_occlusion.PredPreyGrass(cfg, device="cuda:0", **kw),
                                     "wo_los_two_types_seed5", walls=True)


def test_pettingzoo_facades_on_gpu():
    """ParallelEnv driven with the reference's golden action stream returns the reference's observations; the AEC cycle
    runs an episode to exhaustion without leaking a terminated agent."""
    from predpreygrass_amd.env import PredPreyGrass
    from tests.test_env_api import check_aec_runs_to_exhaustion, check_parallel_env_replays_golden
    check_parallel_env_replays_golden(dict(device="cuda:0"))
    check_aec_runs_to_exhaustion(dict(device="cuda:0"))
    assert PredPreyGrass is not None


@pytest.mark.parametrize("f32", [False, True])
def test_packed_observation_image_on_gpu(f32):
    """ppg_pack on the device: 3 sub-batches of 1365/1365/1366 envs -> one image; every section equals plain torch indexing
    of the env tensors; a too small capacity is reported through the header and leaves the row sections alone."""
    import ctypes as C
    from predpreygrass_amd.distributed import parse_image
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    from tests.test_distributed import local_rows_reference
    grp = SubBatchedPredPreyGrass(dict(config_env), batch_size=4096, n_sub=3, device="cuda:0", seed=3)
    grp.reset()
    for _ in range(120):
        grp.step(random_actions=True, auto_reset=True)
    grp.synchronize()
    lib, e0 = grp.subs[0]._lib, grp.subs[0]
    handles = (C.c_void_p * 3)(*[e._handle for e in grp.subs])
    want = [local_rows_reference(e) for e in grp.subs]
    want = {k: torch.cat([w[k] for w in want]) for k in want[0]}
    flags = _abi.PACK_F32 if f32 else 0
    need = int(lib.ppg_pack_bytes(e0._handle, 4096, want["id_pred"].numel(), want["id_prey"].numel(), flags))
    img = torch.zeros(need + 4096, dtype=torch.uint8, device="cuda:0")
    assert lib.ppg_pack(handles, 3, C.c_void_p(img.data_ptr()), img.numel(), flags, e0._stream()) == 0
    torch.cuda.synchronize()
    got = parse_image(img)
    assert got["header"].bytes_used == need and got["header"].n_envs == 4096
    for k, v in want.items():
        v = v.float() if (f32 and k.startswith("obs")) else v
        assert torch.equal(got[k], v), k
    small = torch.full((need // 2,), 7, dtype=torch.uint8, device="cuda:0")
    assert lib.ppg_pack(handles, 3, C.c_void_p(small.data_ptr()), small.numel(), flags, e0._stream()) == 0
    torch.cuda.synchronize()
    hdr = _abi.PpgPackHeader.from_buffer_copy(small[:64].cpu().numpy().tobytes())
    assert hdr.overflow == 1 and hdr.bytes_used == need
    L = _abi.pack_layout(4096, 0, 0, 0, 0, 4)
    assert bool((small[L["id_pred"]:] == 7).all())          # nothing behind the fixed part was touched
    with pytest.raises(OverflowError):
        parse_image(small)


def test_rebalance_changes_scheduling_only_on_gpu():
    """ppg_rebalance (heavy envs are assigned to workgroups first) must not change any result: 2048 envs, 200 calls."""
    envs = [make_env(dict(config_env), 2048), make_env(dict(config_env), 2048)]
    for k, env in enumerate(envs):
        env.reset(seed=77)
        for t in range(200):
            if k == 1 and t % 16 == 0:
                env.rebalance()
            env.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    for n in ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(envs[0], n), getattr(envs[1], n)), n
    assert torch.equal(envs[0].env_state[:, : _abi.ENV_CALLS], envs[1].env_state[:, : _abi.ENV_CALLS])


def test_steps_captured_in_a_hip_graph_replay_bit_identically():
    """ppg_step keeps no host-side state between calls (call counters, RNG keys and row tables live in HBM), so a sequence
    of steps can be captured in a hipGraph (torch.cuda.graph) and replayed: 12 replays of an 8-step graph == 96 eager steps,
    with device-side random actions and with an action buffer that is refilled between replays."""
    cfg = dict(config_env)
    a, b = make_env(cfg, 256), make_env(cfg, 256)
    for env in (a, b):
        env.reset(seed=5)
        env.step(random_actions=True, auto_reset=True)      # first launch (code object load) outside the capture
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(8):
            a.step(random_actions=True, auto_reset=True)
    for _ in range(12):
        graph.replay()
    for _ in range(96):
        b.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    names = ("row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey")
    for n in names:
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(a.env_state[:, : _abi.ENV_CALLS + 1], b.env_state[:, : _abi.ENV_CALLS + 1])

    # explicit actions: the graph holds the POINTER of the action buffer; its contents change between replays
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        a.step(a.actions, auto_reset=True)
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    for _ in range(40):
        acts = torch.randint(0, 9, a.actions.shape, generator=gen, device="cuda:0", dtype=torch.int8)
        a.actions.copy_(acts)
        b.actions.copy_(acts)
        graph2.replay()
        b.step(b.actions, auto_reset=True)
    torch.cuda.synchronize()
    for n in names:
        assert torch.equal(getattr(a, n), getattr(b, n)), n


def test_placement_candidates_change_no_result():
    """SubBatchedPredPreyGrass(placement_candidates=K) steps K candidate buffer sets and keeps the fastest: scheduling only.  After the
    caller's reset() the kept envs are in exactly the state of envs that were never probed."""
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    cfg = {**config_env, "max_steps": 50}
    plain = SubBatchedPredPreyGrass(cfg, batch_size=300, n_sub=3, device="cuda:0", seed=5)
    picked = SubBatchedPredPreyGrass(cfg, batch_size=300, n_sub=3, device="cuda:0", seed=5, placement_candidates=4)
    assert picked.placement_probe_us is not None and len(picked.placement_probe_us) == 4 and plain.placement_probe_us is None
    for g in (plain, picked):
        g.reset()
        for _ in range(70):
            g.step(random_actions=True, auto_reset=True)
        g.synchronize()
    for a, b in zip(plain.subs, picked.subs):
        assert torch.equal(a.env_state[:, : _abi.ENV_CALLS], b.env_state[:, : _abi.ENV_CALLS])
        slot = torch.arange(a.S, device="cuda:0", dtype=torch.int32).unsqueeze(0)
        n_pred = a.env_state[:, _abi.ENV_N_PRED_ROWS:_abi.ENV_N_PRED_ROWS + 1]
        n_prey = a.env_state[:, _abi.ENV_N_PREY_ROWS:_abi.ENV_N_PREY_ROWS + 1]
        live = torch.where(slot < a.pred_capacity, slot < n_pred, slot - a.pred_capacity < n_prey)   # (rows behind the counts keep old bytes)
        assert int(live.sum()) > 300 * 10 // 3
        for n in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "row_cumrew"):
            assert torch.equal(getattr(a, n)[live], getattr(b, n)[live]), n
        assert torch.equal(a.grass_energy, b.grass_energy)
        assert torch.equal(a.obs_pred[live[:, : a.pred_capacity]], b.obs_pred[live[:, : a.pred_capacity]])
        assert torch.equal(a.obs_prey[live[:, a.pred_capacity:]], b.obs_prey[live[:, a.pred_capacity:]])


def test_spread_observation_tensors_change_no_result():
    """obs_spread: the observation tensors on physical pages from ppg_alloc_spread (HIP virtual memory management) -- placement only.
    Same results as on torch's allocator, for the step kernels, the fused rollout and an env that is closed and rebuilt."""
    cfg = {**config_env, "max_steps": 60}
    import ctypes as C
    import gc
    lib = _abi.load_hip_library()

    def spread_stats():
        v = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        assert lib.ppg_spread_stats(*[C.byref(x) for x in v]) == 0
        return [int(x.value) for x in v]   # live bytes, retired ranges, retired bytes
    gc.collect()
    live0, retired0, _ = spread_stats()
    a = BatchedPredPreyGrass(cfg, batch_size=200, device="cuda:0", seed=9)
    b = BatchedPredPreyGrass(cfg, batch_size=200, device="cuda:0", seed=9, obs_spread=4)
    assert b.obs_prey.data_ptr() % (2 << 20) == 0 and b.obs_pred.data_ptr() % (2 << 20) == 0
    assert spread_stats()[0] - live0 >= b.obs_prey.numel() * 8 + b.obs_pred.numel() * 8
    assert float(b.obs_prey.abs().sum()) == 0.0
    for e in (a, b):
        e.set_wave_plan(4, 0, 2)
        e.reset()
        for _ in range(80):
            e.step(random_actions=True, auto_reset=True)
        e.rollout(30, random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    for n in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "grass_energy", "obs_pred", "obs_prey"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    # THE MAPPING FOLLOWS THE TENSORS, NOT THE ENV: an observation tensor (or a view of one) a caller still holds stays readable
    # after close() and after the env object is gone; the pages go back when the last reference does
    keep = b.obs_prey[:3]
    want = keep.clone()
    live1 = spread_stats()[0]
    b.close()
    del b, e          # (`e`: the loop variable above still names the env)
    gc.collect()
    assert torch.equal(keep, want)
    assert spread_stats()[0] < live1                 # obs_pred's pages are back ...
    assert spread_stats()[1] == retired0 + 1
    del keep
    gc.collect()
    assert spread_stats()[1] == retired0 + 2         # ... and now obs_prey's
    c = BatchedPredPreyGrass(cfg, batch_size=64, device="cuda:0", seed=9, obs_spread=2, obs_dtype=torch.bfloat16)
    c.reset()
    c.step(random_actions=True)
    torch.cuda.synchronize()
    assert bool((c.obs_prey.float() != 0).any())
    c.close()


def test_spread_allocator_argument_errors():
    import ctypes as C
    lib = _abi.load_hip_library()
    out = C.c_void_p()
    assert lib.ppg_alloc_spread(0, 0, 4, 1, C.byref(out)) == -1            # PPG_EINVAL: no bytes
    assert lib.ppg_alloc_spread(0, 1 << 20, 0, 1, C.byref(out)) == -1       # spread < 1
    assert lib.ppg_alloc_spread(99, 1 << 20, 2, 1, C.byref(out)) == -4      # PPG_ENODEV
    assert b"device" in lib.ppg_spread_last_error()
    assert lib.ppg_free_spread(C.c_void_p(0x1000)) == -1                    # not ours
    assert lib.ppg_free_spread(None) == 0
    assert lib.ppg_alloc_spread(0, 5 << 20, 3, 7, C.byref(out)) == 0 and out.value and out.value % (2 << 20) == 0
    t = torch.zeros(4, device="cuda:0")   # (the range is ordinary device memory)
    import numpy as np
    holder = type("H", (), {})()
    holder.__cuda_array_interface__ = {"shape": (5 << 20,), "typestr": "|u1", "data": (out.value, False), "version": 2}
    v = torch.as_tensor(holder, device="cuda:0")
    v.fill_(7)
    assert int(v.sum().item()) == 7 * (5 << 20) and float(t.sum()) == 0.0
    del v
    assert lib.ppg_free_spread(out) == 0
    assert lib.ppg_free_spread(out) == -1                                    # twice


def test_spread_allocator_gives_the_memory_back():
    """More bytes than the device has, allocated and freed in turn: ppg_free_spread returns the physical memory (the virtual ranges are
    retired, not re-used)."""
    import ctypes as C
    lib = _abi.load_hip_library()
    total = torch.cuda.get_device_properties(0).total_memory
    size, seen = 8 << 30, set()
    for i in range(int(total * 1.25 / size) + 1):
        out = C.c_void_p()
        assert lib.ppg_alloc_spread(0, size, 1, i, C.byref(out)) == 0, (i, lib.ppg_spread_last_error())
        assert out.value not in seen
        seen.add(out.value)
        assert lib.ppg_free_spread(out) == 0
    # what that costs: virtual address space only -- counted, so that a long-lived process can see it (2^47 bytes exist)
    live, ranges, nbytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert lib.ppg_spread_stats(C.byref(live), C.byref(ranges), C.byref(nbytes)) == 0
    assert ranges.value >= len(seen) and nbytes.value >= len(seen) * size
