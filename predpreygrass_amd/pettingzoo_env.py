"""PettingZoo-style façades over `PredPreyGrass`.

The reference declares pettingzoo as a dependency but never imports it (its README points to a
different repository for the PettingZoo version), so there is NO reference behaviour to pin these
against: API shape only, parity unpinned.  They follow the PettingZoo 1.24
conventions: `agents` lists the agents that are still acting, the per-agent dicts carry no
"__all__" key, dead agents leave `agents` after the step that reports their termination.
"""
from __future__ import annotations

from .env import PredPreyGrass

try:
    from pettingzoo import ParallelEnv as _ParallelBase  # type: ignore
    from pettingzoo import AECEnv as _AECBase  # type: ignore
except Exception:
    class _ParallelBase:
        pass

    class _AECBase:
        pass


class PredPreyGrassParallelEnv(_ParallelBase):
    metadata = {"name": "predpreygrass_amd_v0", "render_modes": []}

    def __init__(self, config=None, env_class=None, **kw):
        # env_class: the dict env to wrap (default: the base env; red_queen / walls_occlusion / drive_conditioned .PredPreyGrass)
        self._env = (env_class or PredPreyGrass)(config, **kw)
        self.possible_agents = list(self._env.possible_agents)
        self.agents = []

    def observation_space(self, agent):
        return self._env.observation_spaces[agent]

    def action_space(self, agent):
        return self._env.action_spaces[agent]

    def reset(self, seed=None, options=None):
        obs, _ = self._env.reset(seed=seed, options=options)
        self.agents = list(obs)
        return obs, {a: {} for a in obs}

    def step(self, actions):
        obs, rew, term, trunc, infos = self._env.step({a: actions[a] for a in self.agents if a in actions})
        rew = {k: rew[k] for k in obs}   # (the walls env also reports agents that are gone; PettingZoo dicts follow obs)
        term = {k: term[k] for k in obs}
        trunc = {k: trunc[k] for k in obs}
        self.agents = [a for a in obs if not term[a] and not trunc[a]]
        return obs, rew, term, trunc, {a: dict(infos.get(a, {})) for a in obs}

    def state(self):
        return self._env.grid_world_state

    def close(self):
        self._env.close()

    @property
    def unwrapped(self):
        return self

    # (pettingzoo's base classes provide these; spelled out because the package may be absent)
    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    @property
    def observation_spaces(self):
        return self._env.observation_spaces

    @property
    def action_spaces(self):
        return self._env.action_spaces


class PredPreyGrassAECEnv(_AECBase):
    """Agent-environment-cycle view: actions are buffered per agent; the underlying parallel step
    runs when the last live agent of the cycle has acted (PettingZoo's parallel_to_aec scheme)."""
    metadata = {"name": "predpreygrass_amd_aec_v0", "render_modes": [], "is_parallelizable": True}

    def __init__(self, config=None, **kw):
        self._par = PredPreyGrassParallelEnv(config, **kw)
        self.possible_agents = self._par.possible_agents
        self.agents = []

    def observation_space(self, agent):
        return self._par.observation_space(agent)

    def action_space(self, agent):
        return self._par.action_space(agent)

    def reset(self, seed=None, options=None):
        obs, infos = self._par.reset(seed=seed, options=options)
        self.agents = list(self._par.agents)
        self._obs = dict(obs)
        self.rewards = {a: 0.0 for a in self.agents}
        self._cumulative_rewards = {a: 0.0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}
        self._pending = {}
        self._order = list(self.agents)
        self._cursor = 0
        self.agent_selection = self._order[0] if self._order else None

    def observe(self, agent):
        return self._obs[agent]

    def last(self, observe=True):
        a = self.agent_selection
        return (self._obs[a] if observe else None, self._cumulative_rewards[a], self.terminations[a],
                self.truncations[a], self.infos[a])

    def agent_iter(self, max_iter=2 ** 63):
        n = 0
        while self.agents and n < max_iter:
            yield self.agent_selection
            n += 1

    def step(self, action):
        a = self.agent_selection
        if self.terminations[a] or self.truncations[a]:
            # dead-step: the agent is removed (PettingZoo's _was_dead_step); action must be None.  The cursor stays: the
            # next agent of the cycle has moved into this position.
            if action is not None:
                raise ValueError("when an agent is dead, the only valid action is None")
            self.agents.remove(a)
            for d in (self.rewards, self._cumulative_rewards, self.terminations, self.truncations, self.infos):
                d.pop(a, None)
            del self._order[self._cursor]
        else:
            self._cumulative_rewards[a] = 0.0
            self._pending[a] = action
            self._cursor += 1
        if self._cursor >= len(self._order) and self._order:
            # the cycle is complete: every live agent has acted and every agent that was reported dead has been
            # dead-stepped, so no terminated agent can be left behind in `agents` when the order is rebuilt
            obs, rew, term, trunc, infos = self._par.step(self._pending)
            self._pending = {}
            for k in obs:
                if k not in self.agents:
                    self.agents.append(k)
                self._obs[k] = obs[k]
                self.rewards[k] = rew[k]
                self._cumulative_rewards[k] = self._cumulative_rewards.get(k, 0.0) + rew[k]
                self.terminations[k], self.truncations[k], self.infos[k] = term[k], trunc[k], infos[k]
            self._order = list(obs)
            self._cursor = 0
        self.agent_selection = self._order[self._cursor] if self._order else None

    def close(self):
        self._par.close()

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    @property
    def unwrapped(self):
        return self


def parallel_env(config=None, **kw):
    return PredPreyGrassParallelEnv(config, **kw)


def env(config=None, **kw):
    return PredPreyGrassAECEnv(config, **kw)
