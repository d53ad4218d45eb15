"""The second-generation CPU oracle (oracle/rq_oracle.c) against golden vectors produced by the reference itself
(red_queen/predpreygrass_rllib_env.py).  This is what pins the oracle; everything else is checked against it."""
import numpy as np
import pytest

from oracle.rq_oracle import RQOracleEnv, philox_uniform
from tests.golden_io_rq import RQGoldenCase, call_digest, case_names

CASES = case_names() + case_names(walls=True)


def test_golden_cases_present():
    assert len(case_names()) >= 7 and len(case_names(walls=True)) >= 5


@pytest.mark.parametrize("name", CASES)
def test_oracle_replays_reference(name):
    case = RQGoldenCase(name)
    env = RQOracleEnv(case.config, walls=case.walls)
    if case.walls:
        env.set_walls(case.wall_xy)
    obs, info = env.reset_from_placement(*case.placement)
    assert info == {}
    ref = case.reset_obs()
    assert list(obs) == case.reset_keys
    for k in ref:
        assert obs[k].dtype == np.float32 and np.array_equal(obs[k], ref[k])
    for t in range(case.n_calls):
        u, n_used = case.uniforms(t, extra=3)
        obs, rew, term, trunc, infos = env.step(case.actions(t), uniforms=u)
        assert infos == case.infos(t), t
        assert env.last_draws == n_used, (t, env.last_draws, n_used)
        assert env.last_fallback_spawns == 0 and env.last_failed_spawns == 0
        recs = case.records(t)
        assert list(obs) == [r[0] for r in recs], t
        extra = case.extras(t)
        assert list(rew) == list(obs) + extra and list(term)[:-1] == list(rew) and list(trunc)[:-1] == list(rew)
        assert all(rew[a] == 0.0 and term[a] is False and trunc[a] is False for a in extra)
        for name_, r, te, tr in recs:
            assert rew[name_] == r, (t, name_)
            assert term[name_] == te and trunc[name_] == tr, (t, name_)
        assert (term["__all__"], trunc["__all__"]) == case.flags(t)
        grid = env.grid_world_state
        assert grid.dtype == np.float32
        assert call_digest(grid, obs, rew, term, trunc, sort_scalars=case.walls) == case.digest(t), t
        assert env.agents == case.agents_after[t], t
        full = case.full(t)
        if full is not None:
            fobs, fgrid, state, grass, next_idx = full
            assert np.array_equal(grid, fgrid)
            for k in fobs:
                assert np.array_equal(obs[k], fobs[k]), (t, k)
            for a, s in state.items():
                assert env.agent_state(a) == s, (t, a)
            assert np.array_equal(env.grass_state()[1], grass)
            assert env.next_ids == next_idx


def test_philox_uniform_is_a_53_bit_fraction():
    for d in range(64):
        u = philox_uniform(1234, 2, 17, d)
        assert 0.0 <= u < 1.0 and (u * 2.0 ** 53) == int(u * 2.0 ** 53)
    assert len({philox_uniform(1234, 2, 17, d) for d in range(64)}) == 64


def test_oracle_random_rollout_is_deterministic():
    case = RQGoldenCase("rq_mixed_types_seed7")
    a, b = RQOracleEnv(case.config), RQOracleEnv(case.config)
    assert a.rollout_random(99, 300) == 300 and b.rollout_random(99, 300) == 300
    assert a.last_records() == b.last_records()
    assert a.agents == b.agents and np.array_equal(a.grid_world_state, b.grid_world_state)
