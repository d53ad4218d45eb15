"""PettingZoo conformance of the two façades, restated as own tests.

The `pettingzoo` package is not installed here (and the reference never imports it: SURVEY.md 0.2), so its `parallel_api_test` /
`api_test` / `seed_test` cannot be run.  What those tests demand of an environment -- the invariants below, written down from the
PettingZoo 1.24 API documentation -- is checked directly, on the emulated kernel, for an env WITH agent generation (newborns
join `agents`) and agent death:

  parallel: reset() -> (observations, infos) keyed exactly by `agents`; step(actions of the live agents) -> five dicts with one
    entry for every agent that acted (and for agents born in the step); terminated / truncated agents leave `agents` after the
    step that reports them and never return; `agents` is always a subset of `possible_agents`; spaces are functions of the agent
    that return the SAME object on every call; observations have the space's shape and dtype; rewards are numbers, flags bools,
    infos dicts; num_agents / max_num_agents; the episode ends with `agents == []`.
  AEC: after reset `agent_selection` is in `agents` and rewards / terminations / truncations / infos / _cumulative_rewards are keyed
    by `agents`; last() reports the selected agent's observation and its reward accumulated since it last acted; a live agent acts
    once per cycle in `agents` order; a dead agent must be stepped with None (anything else raises ValueError), is removed by that
    step and its entries vanish; agent_iter() ends when `agents` is empty.
  seeding: the same seed and the same actions give the same episode, another seed another one.
"""
import numbers

import numpy as np
import pytest

from predpreygrass_amd.config import config_env
from predpreygrass_amd.pettingzoo_env import PredPreyGrassAECEnv, PredPreyGrassParallelEnv
from tests.emu_backend import library

CFG = {**config_env, "max_steps": 60, "n_initial_active_predator": 5, "n_initial_active_prey": 9, "initial_num_grass": 40,
       "energy_gain_per_step_grass": 0.3}          # births within a few steps, deaths throughout, truncation at 60


def check_spaces(env):
    assert isinstance(env.possible_agents, list) and len(env.possible_agents) > 0
    assert env.max_num_agents == len(env.possible_agents)
    for a in env.possible_agents[:3] + env.possible_agents[-3:]:
        assert env.observation_space(a) is env.observation_space(a)
        assert env.action_space(a) is env.action_space(a)
        assert env.action_space(a).n == 9


def check_observation(env, agent, obs):
    space = env.observation_space(agent)
    assert isinstance(obs, np.ndarray) and obs.shape == space.shape and obs.dtype == space.dtype
    assert np.isfinite(obs).all()
    # (values: the reference declares Box(0, 100) (its predpreygrass_rllib_env.py:84-94) while a dying agent's cell can show an energy
    # just below zero for one step -- the declaration is mirrored, the values are the reference's)


def test_parallel_api_invariants():
    env = PredPreyGrassParallelEnv(CFG, _library=library())
    check_spaces(env)
    rng = np.random.default_rng(0)
    obs, infos = env.reset(seed=3)
    assert set(obs) == set(infos) == set(env.agents) and len(env.agents) == 14 and env.num_agents == 14
    finished, born, steps = set(), 0, 0
    while env.agents:
        live = list(env.agents)
        assert set(live) <= set(env.possible_agents) and not (set(live) & finished)
        actions = {a: int(rng.integers(0, 9)) for a in live}
        obs, rew, term, trunc, infos = env.step(actions)
        steps += 1
        keys = set(obs)
        assert keys == set(rew) == set(term) == set(trunc) == set(infos)
        assert set(live) <= keys                                     # everybody who acted is reported, dead or alive
        born += len(keys - set(live))
        assert not ((keys - set(live)) & finished)                   # newcomers are new
        for a in keys:
            check_observation(env, a, obs[a])
            assert isinstance(rew[a], numbers.Real) and isinstance(term[a], (bool, np.bool_)) and isinstance(trunc[a], (bool, np.bool_))
            assert isinstance(infos[a], dict)
            if term[a] or trunc[a]:
                finished.add(a)
        assert set(env.agents) == {a for a in keys if not term[a] and not trunc[a]}
        assert env.num_agents == len(env.agents)
        assert steps <= CFG["max_steps"] + 1
    assert env.agents == [] and born > 0 and len(finished) >= 14
    # a fresh episode after the end
    obs, _ = env.reset(seed=4)
    assert len(env.agents) == 14 and set(obs) == set(env.agents)
    env.close()


def run_aec_episode(seed, action_seed, max_iter=20000):
    env = PredPreyGrassAECEnv(CFG, _library=library())
    env.reset(seed=seed)
    rng = np.random.default_rng(action_seed)
    trace = []
    for agent in env.agent_iter(max_iter=max_iter):
        o, r, te, tr, info = env.last()
        trace.append((agent, o.tobytes(), float(r), bool(te), bool(tr)))
        env.step(None if (te or tr) else int(rng.integers(0, 9)))
    assert env.agents == []
    env.close()
    return trace


def test_aec_api_invariants():
    env = PredPreyGrassAECEnv(CFG, _library=library())
    check_spaces(env)
    env.reset(seed=3)
    rng = np.random.default_rng(1)
    assert env.agent_selection in env.agents and env.num_agents == 14
    for d in (env.rewards, env.terminations, env.truncations, env.infos, env._cumulative_rewards):
        assert set(d) == set(env.agents)
    assert all(v == 0 for v in env._cumulative_rewards.values())
    acted_this_cycle, removed, dead_steps, cycles = [], set(), 0, 0
    since_acted = {a: 0.0 for a in env.agents}      # what last() must report: rewards received since the agent last acted
    for agent in env.agent_iter(max_iter=50000):
        assert agent == env.agent_selection and agent in env.agents and agent not in removed
        o, r, te, tr, info = env.last()
        check_observation(env, agent, o)
        assert o.tobytes() == env.observe(agent).tobytes()
        assert r == pytest.approx(since_acted.get(agent, 0.0)) and isinstance(info, dict)
        if te or tr:
            with pytest.raises(ValueError):
                env.step(3)                              # a dead agent takes None only
            env.step(None)
            dead_steps += 1
            removed.add(agent)
            assert agent not in env.agents
            for d in (env.rewards, env.terminations, env.truncations, env.infos, env._cumulative_rewards):
                assert agent not in d
            continue
        assert agent not in acted_this_cycle            # once per cycle
        acted_this_cycle.append(agent)
        env.step(int(rng.integers(0, 9)))
        since_acted[agent] = 0.0
        # the underlying step ran iff the cursor wrapped: then every agent's pending reward grew by this step's reward
        if env._cursor == 0 and env._pending == {}:
            cycles += 1
            for a in env.agents:
                since_acted[a] = since_acted.get(a, 0.0) + float(env.rewards.get(a, 0.0))
            acted_this_cycle = []
            for d in (env.rewards, env.terminations, env.truncations, env.infos, env._cumulative_rewards):
                assert set(d) == set(env.agents)
            assert set(env.agents) <= set(env.possible_agents)
    assert env.agents == [] and env.agent_selection is None
    assert dead_steps >= 14 and cycles >= 10
    env.close()


def test_same_seed_same_episode_other_seed_other_episode():
    a = run_aec_episode(7, 70)
    b = run_aec_episode(7, 70)
    c = run_aec_episode(8, 70)
    assert a == b
    assert a != c
    # the parallel façade likewise
    def par(seed):
        env = PredPreyGrassParallelEnv(CFG, _library=library())
        rng = np.random.default_rng(5)
        obs, _ = env.reset(seed=seed)
        out = [sorted((k, v.tobytes()) for k, v in obs.items())]
        for _ in range(25):
            if not env.agents:
                break
            o, r, te, tr, _ = env.step({a: int(rng.integers(0, 9)) for a in env.agents})
            out.append(sorted((k, v.tobytes(), float(r[k]), bool(te[k])) for k, v in o.items()))
        env.close()
        return out
    assert par(11) == par(11) and par(11) != par(12)
