"""The per-agent bookkeeping the second-generation reference env keeps beside its transition -- ``unique_agents``,
``unique_agent_stats``, ``death_agents_stats``, ``per_step_agent_data``, ``agent_ages``, ``agent_parents``,
``agent_offspring_counts``, ``agent_live_offspring_ids``, ``agent_activation_counts`` -- rebuilt on the host from what a
``step()`` of the dict class fetches anyway (the row tables before and after the call).  None of it feeds back into the
transition; the reference's evaluation scripts read it (red_queen/predpreygrass_rllib_env.py, "RQ": 99-116, 236-245, 272-294,
502, 529-533, 560-570, 608-633, 675-687, 739-745, 825-857, 987-1017; the walls variant, "WO", keeps the same books with
two differences noted below).

Why on the host: every figure is a Python float accumulated in the reference's own order (``+=`` of float64 values), a
per-agent sum over the agent's life.  The transition's inputs to them -- who moved where, who ate, who died, who was born to
whom -- are all in the tables; the arithmetic (decay, distance * factor * energy, capped gains) is restated here operation by
operation, so the sums are bit-identical, and every energy the mirror derives is cross-checked against the device's
(``check=True``: the tests).  Cost: one pass over the live agents per call, only when the env was built with
``analytics=True`` (the default of the dict classes, like the reference); the batched tensor API never pays for it.

The reference's quirks are kept: in RQ a predator that catches a prey gets ``death_step`` / ``death_cause = "eaten"`` /
``final_energy`` written into ITS OWN stats and is entered into ``death_agents_stats`` (RQ:623-633 use the predator's uid; WO
writes them into the prey's stats, with the predator's energy as ``final_energy``, WO:879-888);
the entry of an agent that really dies is a copy with ``lifetime`` and ``parent`` added (RQ:239-244); the lists in
``per_step_agent_data[...]["offspring_ids"]`` are the live lists, not copies (RQ:290).
"""
from __future__ import annotations

import math

import numpy as np


ZERO_DELTAS = {"decay": 0.0, "move": 0.0, "eat": 0.0, "repro": 0.0}


class AgentAnalytics:
    def __init__(self, cfg, walls=False, check=False):
        self.cfg = cfg
        self.walls = bool(walls)
        self.check = bool(check)
        self.clear(possible_agents=())

    # -- RQ:99-116 ---------------------------------------------------------------------------------------------
    def clear(self, possible_agents):
        self.agent_ages = {}
        self.agent_parents = {}
        self.unique_agents = {}
        self.unique_agent_stats = {}
        self.per_step_agent_data = []
        self._per_agent_step_deltas = {}
        self.agent_offspring_counts = {}
        self.agent_live_offspring_ids = {}
        self.agent_activation_counts = {a: 0 for a in possible_agents}
        self.death_agents_stats = {}
        self.death_cause_prey = {}
        self._energy = {}   # the mirror's own agent_energies (cross-checked against the device's)

    # -- RQ:987-1017 -------------------------------------------------------------------------------------------
    def register(self, agent_id, current_step, parent_unique_id=None):
        reuse_index = self.agent_activation_counts.get(agent_id, 0)
        unique_id = f"{agent_id}_{reuse_index}"
        self.unique_agents[agent_id] = unique_id
        self.agent_activation_counts[agent_id] = reuse_index + 1
        self.agent_ages[agent_id] = 0
        self.agent_offspring_counts[agent_id] = 0
        self.agent_live_offspring_ids[agent_id] = []
        self.agent_parents[agent_id] = parent_unique_id
        self.unique_agent_stats[unique_id] = self._fresh_stats(agent_id, current_step, parent_unique_id)

    @staticmethod
    def _fresh_stats(agent_id, current_step, parent_unique_id):
        return {
            "birth_step": current_step,
            "parent": parent_unique_id,
            "offspring_count": 0,
            "distance_traveled": 0.0,
            "times_ate": 0,
            "energy_gained": 0.0,
            "energy_spent": 0.0,
            "avg_energy_sum": 0.0,
            "avg_energy_steps": 0,
            "cumulative_reward": 0.0,
            "policy_group": "_".join(agent_id.split("_")[:3]),
            "mutated": False,
            "death_step": None,
            "death_cause": None,
            "final_energy": None,
            "avg_energy": None,
        }

    def reset(self, possible_agents, agents, energies):
        """RQ:119-133 (_init_reset_variables registers the initial agents at current_step 0) and RQ:168-180 (energies)."""
        self.clear(possible_agents)
        for a in agents:
            self.register(a, 0)
        self._energy = dict(energies)

    def _type_specific(self, key, agent_id):
        """RQ:1099-1106."""
        raw = self.cfg.get(key, 0.0)
        if isinstance(raw, dict):
            for k in raw:
                if agent_id.startswith(k):
                    return raw[k]
            raise KeyError(f"Type-specific key '{agent_id}' not found under '{key}'")
        return raw

    def _death_stat(self, agent, current_step, cause, cum_reward, energy_of=None):
        """RQ:560-570 / 623-633: the block both death paths share.  For a catch RQ applies it to the PREDATOR (RQ:624 takes the
        predator's uid); WO applies it to the caught prey but still records the predator's energy as final_energy (WO:879-888)."""
        uid = self.unique_agents[agent]
        stat = self.unique_agent_stats[uid]
        stat["death_step"] = current_step
        stat["death_cause"] = cause
        stat["final_energy"] = self._energy[agent if energy_of is None else energy_of]
        steps = max(stat["avg_energy_steps"], 1)
        stat["avg_energy"] = stat["avg_energy_sum"] / steps
        stat["cumulative_reward"] = cum_reward
        self.death_agents_stats[uid] = stat

    # ----------------------------------------------------------------------------------------------------------
    def step(self, *, current_step, action_names, agents_in_order, insertion_order, pos_before, pos_after, cum_before,
             grass_pos, grass_energy_before, terminated, ate, newborn, lastrep_after, energy_after, grass_energy_after=None):
        """One non-truncated call (RQ:213-294).

        current_step     the env's counter before the call
        action_names     keys of the action dict in its order (RQ:466,502,521 walk it)
        agents_in_order  self.agents during the call: survivors of the previous call in the reference's order, no newborns
        insertion_order  names in ``agent_positions`` insertion order at the start of the call (RQ:585 / 653 search in it)
        pos_before / pos_after   name -> (x, y); cum_before: name -> cumulative_rewards before the call
        grass_pos, grass_energy_before   patch lists (index = k of "grass_k"), energies before the regrowth of this call
        terminated / ate   sets of names (device flags); newborn: names born in this call, in birth order
        lastrep_after    name -> the device's agent_last_reproduction after the call (== current_step: a parent of this call)
        energy_after     name -> the device's agent_energies after the call (incl. the agents that died in it): who reproduced,
                         and the cross-check of the mirror's own energies
        """
        cfg = self.cfg
        E = self._energy
        live = set(pos_before)
        deltas = self._per_agent_step_deltas
        # step 1: decay (RQ:466-489)
        for a in action_names:
            if a not in live:
                continue
            decay = cfg["energy_loss_per_step_predator"] if "predator" in a else cfg["energy_loss_per_step_prey"]
            E[a] -= decay
            deltas[a] = {"decay": -decay, "move": 0.0, "eat": 0.0, "repro": 0.0}
        # step 2: ages (RQ:497-502: every key of the action dict)
        for a in action_names:
            if a in self.agent_ages:
                self.agent_ages[a] += 1
        # step 3: grass (RQ:504-515)
        gcap = cfg.get("max_energy_grass", float("inf"))
        gain_g = cfg["energy_gain_per_step_grass"]
        G = [min(g + gain_g, gcap) for g in grass_energy_before]
        # step 4: movement (RQ:517-545; the positions are the device's)
        factor = cfg.get("move_energy_cost_factor", 0.01)
        for a in action_names:
            if a not in live:
                continue
            old, new = pos_before[a], pos_after[a]
            distance = math.sqrt((new[0] - old[0]) ** 2 + (new[1] - old[1]) ** 2)      # RQ:310
            move_cost = distance * factor * E[a]                                        # RQ:312
            E[a] -= move_cost
            deltas[a]["move"] = -move_cost
            st = self.unique_agent_stats[self.unique_agents[a]]
            # RQ:530 adds np.linalg.norm(new - old) of two integer vectors: the correctly rounded square root of the same exact integer
            # as RQ:310's math.sqrt -- the same float64, as a numpy scalar like the reference's (a tenth of the norm call's cost)
            st["distance_traveled"] += np.float64(distance)
            st["energy_spent"] += move_cost
            st["avg_energy_sum"] += E[a]
            st["avg_energy_steps"] += 1
        # step 5: engagements in self.agents order (RQ:225-233)
        present = set(live)
        grass_at = {}
        for k, p in enumerate(grass_pos):
            grass_at.setdefault(tuple(p), k)                    # RQ:653: the first patch on the cell
        cum = dict(cum_before)
        eff = cfg.get("energy_transfer_efficiency", 1.0)
        for a in agents_in_order:
            if a not in present:
                continue
            if E[a] <= 0:                                       # RQ:547-580
                self._death_stat(a, current_step, "starved", cum.get(a, 0.0))
                present.discard(a)
                assert a in terminated, (a, "the device kept an agent the mirror starves")
                continue
            stat = self.unique_agent_stats[self.unique_agents[a]]
            if "predator" in a:                                 # RQ:582-645
                here = pos_after[a]
                caught = next((q for q in insertion_order if q in present and "prey" in q and pos_after[q] == here), None)
                assert (caught is not None) == (a in ate), (a, caught, "catch disagrees with the device's flag")
                if caught is not None:
                    r = self._type_specific("reward_predator_catch_prey", a)
                    cum[a] = cum.get(a, 0) + r
                    gain = min(E[caught], cfg.get("max_energy_gain_per_prey", float("inf"))) * eff
                    E[a] += gain
                    deltas.setdefault(a, dict(ZERO_DELTAS))["eat"] = gain   # (an agent that did not act has no entry: the reference fails here)
                    E[a] = min(E[a], cfg.get("max_energy_predator", float("inf")))
                    stat["times_ate"] += 1
                    stat["energy_gained"] += E[caught]
                    stat["cumulative_reward"] += r
                    cum[caught] = cum.get(caught, 0.0) + self._type_specific("penalty_prey_caught", caught)
                    if self.walls:
                        self._death_stat(caught, current_step, "eaten", cum.get(caught, 0.0), energy_of=a)   # WO:879-888
                    else:
                        self._death_stat(a, current_step, "eaten", cum.get(a, 0.0))   # RQ:623-633: the predator's uid (sic)
                    present.discard(caught)
                    assert caught in terminated, (caught, "the device kept a prey the mirror has caught")
                else:
                    r = self._type_specific("reward_predator_step", a)
                cum[a] = cum.get(a, 0) + r
            else:                                               # RQ:647-693
                k = grass_at.get(tuple(pos_after[a]))
                assert (k is not None) == (a in ate), (a, k, "grazing disagrees with the device's flag")
                if k is not None:
                    r = self._type_specific("reward_prey_eat_grass", a)
                    cum[a] = cum.get(a, 0) + r
                    gain = min(G[k], cfg.get("max_energy_gain_per_grass", float("inf"))) * eff
                    E[a] += gain
                    deltas.setdefault(a, dict(ZERO_DELTAS))["eat"] = gain
                    E[a] = min(E[a], cfg.get("max_energy_prey", float("inf")))
                    stat["times_ate"] += 1
                    stat["energy_gained"] += G[k]
                    stat["cumulative_reward"] += r
                    G[k] = 0
                else:
                    r = self._type_specific("reward_prey_step", a)
                    stat["cumulative_reward"] += r
                cum[a] = cum.get(a, 0) + r
        # step 6: removals (RQ:236-245)
        pending = [a for a in agents_in_order if a in terminated]
        assert set(pending) == live - present, (sorted(pending), sorted(live - present))
        for a in pending:
            uid = self.unique_agents[a]
            self.death_agents_stats[uid] = {**self.unique_agent_stats[uid], "lifetime": self.agent_ages[a],
                                            "parent": self.agent_parents[a]}
            del self.unique_agents[a]
        # step 7: reproduction in self.agents order (RQ:248-254): the k-th child of a species belongs to the k-th parent of
        # that species whose agent_last_reproduction the call set (a parent has at most one child per call)
        kids = {"predator": [n for n in newborn if "predator" in n], "prey": [n for n in newborn if "prey" in n]}
        for a in agents_in_order:
            if a in pending or lastrep_after.get(a) != current_step:
                continue
            sp = "predator" if "predator" in a else "prey"
            e0 = cfg[f"initial_energy_{sp}"]
            # (agent_last_reproduction starts at -cooldown, which IS the current step for cooldown 0 at step 0: a parent also paid e0)
            if not kids[sp] or np.float64(E[a] - e0).tobytes() != np.float64(energy_after[a]).tobytes():
                continue
            child = kids[sp].pop(0)
            deltas[child] = {"decay": 0.0, "move": 0.0, "eat": 0.0, "repro": 0.0}
            self.register(child, current_step, parent_unique_id=self.unique_agents[a])
            child_uid = self.unique_agents[child]
            self.agent_live_offspring_ids[a].append(child_uid)
            self.agent_offspring_counts[a] += 1
            self.unique_agent_stats[child_uid]["mutated"] = a.split("_")[1] != child.split("_")[1]   # RQ:708-712
            self.unique_agent_stats[self.unique_agents[a]]["offspring_count"] += 1
            E[child] = e0 * cfg.get("reproduction_energy_efficiency", 1.0)
            E[a] -= e0
            deltas.setdefault(a, dict(ZERO_DELTAS))["repro"] = -e0
            if sp == "prey" or self.walls:                      # RQ:856-857 (prey only); WO:1026-1027 adds the predators
                self.unique_agent_stats[self.unique_agents[a]]["cumulative_reward"] += self._type_specific(f"reproduction_reward_{sp}", a)
        assert not kids["predator"] and not kids["prey"], ("newborns without a parent", kids)
        for a in pending:
            del E[a]
        if self.check:
            for a, e in E.items():
                assert np.float64(e).tobytes() == np.float64(energy_after[a]).tobytes(), (a, e, energy_after[a])
            if grass_energy_after is not None:
                assert [float(g) for g in G] == [float(g) for g in grass_energy_after], "grass energies"
        return pending

    def record_step(self, sorted_agents, pending, positions):
        """RQ:272-294: after ``self.agents.sort()``, one entry per agent that is not pending removal."""
        step_data = {}
        for a in sorted_agents:
            if a in pending:
                continue
            if self.walls:   # WO:406
                d = self._per_agent_step_deltas.get(a, {"decay": 0.0, "move": 0.0, "eat": 0.0, "repro": 0.0})
            else:
                d = self._per_agent_step_deltas[a]
            step_data[a] = {
                "position": positions[a],
                "energy": self._energy[a],
                "energy_decay": d["decay"],
                "energy_movement": d["move"],
                "energy_eating": d["eat"],
                "energy_reproduction": d["repro"],
                "age": self.agent_ages[a],
                "offspring_count": self.agent_offspring_counts[a],
                "offspring_ids": self.agent_live_offspring_ids[a],
            }
        self.per_step_agent_data.append(step_data)
        self._per_agent_step_deltas.clear()

    # -- snapshot (RQ:893-939) ---------------------------------------------------------------------------------
    def snapshot(self):
        return {"unique_agents": self.unique_agents.copy(), "agent_activation_counts": self.agent_activation_counts.copy(),
                "agent_ages": self.agent_ages.copy(), "death_cause_prey": self.death_cause_prey.copy(),
                "per_step_agent_data": self.per_step_agent_data.copy(), "_analytics_energy": dict(self._energy)}

    def restore(self, snap):
        self.unique_agents = snap["unique_agents"].copy()
        self.agent_activation_counts = snap["agent_activation_counts"].copy()
        self.agent_ages = snap["agent_ages"].copy()
        self.death_cause_prey = snap["death_cause_prey"].copy()
        self.per_step_agent_data = snap["per_step_agent_data"].copy()
        self._energy = dict(snap["_analytics_energy"])
        # The books the reference does NOT snapshot (unique_agent_stats, parents, offspring lists: RQ:906-913) stay as they are.  After
        # a reset() in between they no longer know the restored agents -- the reference then fails at its next step (RQ:530); here
        # such agents get fresh entries so that a restored env keeps stepping.
        for agent_id, uid in self.unique_agents.items():
            if uid not in self.unique_agent_stats:
                self.unique_agent_stats[uid] = self._fresh_stats(agent_id, None, None)
            self.agent_parents.setdefault(agent_id, None)
            self.agent_offspring_counts.setdefault(agent_id, 0)
            self.agent_live_offspring_ids.setdefault(agent_id, [])
            self.agent_ages.setdefault(agent_id, 0)
