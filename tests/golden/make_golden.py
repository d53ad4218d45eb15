#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the produced ``*.npz``
files are data (inputs + expected outputs) and are what travels to the GPU box.

    python tests/golden/make_golden.py            # regenerate everything

For every case the reference env
(/root/reference/predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py,
imported through oracle/ref_shim.py) is reset with a seed, its initial placement
is captured (reset placement depends on numpy's PCG64 + CPython set order and is
treated as captured input), then driven with the RLlib-style
live-agent protocol: actions only for agents that got
an observation last call and were not terminated, drawn one scalar at a time
from ``np.random.default_rng(action_seed).integers(0, 9)`` in dict order.

Stored per case:
  config_json          the config overrides (on top of config_env.py defaults)
  pred_xy/prey_xy/grass_xy   captured placement, id order
  act_off[T+1], act_type/act_id/act_val      action dicts, flattened, dict order
  rec_off[T+1], rec_type/rec_id/rec_reward/rec_term/rec_trunc   returned dicts, dict order
  term_all[T], trunc_all[T]
  digest[T,32]         sha256 per call (definition in `call_digest`)
  survey_rolling16     SURVEY.md Appendix-B rolling digest, hex, at the last call
  agents_after[T] (json)  env.agents after each call (the sorted list)
  full_calls[], obs_off[], obs_data     complete observation tensors for the
                       calls listed in full_calls (dict order, concatenated)
  grid_calls == full_calls, grid_data   complete (4,G,G) grid after those calls
  reset_obs_data       observations returned by reset (dict order, concatenated)
  state_calls[], st_* : agent positions/energies/cumulative rewards for those calls
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.ref_shim import make_reference_env, reference_default_config  # noqa: E402

PREDATOR, PREY = 0, 1


def parse_agent(name):
    kind, idx = name.rsplit("_", 1)
    return (PREDATOR if kind == "predator" else PREY), int(idx)


def call_digest(grid, obs, rew, term, trunc) -> bytes:
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(grid, dtype=np.float64).tobytes())
    for k, v in obs.items():
        h.update(k.encode() + np.ascontiguousarray(v, dtype=np.float64).tobytes())
    for k, v in rew.items():
        h.update(k.encode() + np.float64(v).tobytes())
    for k, v in term.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    for k, v in trunc.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    return h.digest()


CASES = {
    # name: (overrides, reset seed, action seed, shuffle action order, full-every)
    "c1_seed0": ({"n_initial_active_predator": 4, "n_initial_active_prey": 8, "initial_num_grass": 30}, 0, 0, False, 1),
    "default_seed0": ({}, 0, 0, False, 40),
    "default_seed1": ({}, 1, 1001, False, 25),
    "c4_seed0": ({"grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32,
                  "predator_obs_range": 7, "prey_obs_range": 7}, 0, 1000, False, 6),
    "dense_seed0": ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20,
                     "initial_num_grass": 25, "predator_obs_range": 5, "prey_obs_range": 7,
                     "max_steps": 150}, 0, 1000, False, 10),
    "dense_seed3": ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20,
                     "initial_num_grass": 25, "predator_obs_range": 5, "prey_obs_range": 7,
                     "max_steps": 150}, 3, 1003, False, 6),
    "dense_shuffled_seed17": ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20,
                               "initial_num_grass": 25, "predator_obs_range": 5, "prey_obs_range": 7,
                               "max_steps": 150}, 17, 1017, True, 10),
    "rewards_seed3": ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20,
                       "initial_num_grass": 25, "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 150,
                       "reward_predator_catch_prey": 1.5, "reward_prey_eat_grass": 0.25,
                       "reward_predator_step": -0.01, "reward_prey_step": 0.02, "penalty_prey_caught": -2.0,
                       "reproduction_reward_predator": 7.0, "reproduction_reward_prey": 3.0}, 3, 1003, False, 8),
    "pool_seed3": ({"n_possible_predators": 8, "n_possible_prey": 12, "energy_gain_per_step_grass": 0.2,
                    "max_steps": 300}, 3, 1003, False, 15),
    "even_obs_seed0": ({"predator_obs_range": 6, "prey_obs_range": 8, "grid_size": 12, "initial_num_grass": 40,
                        "max_steps": 200}, 0, 1000, False, 20),
    # base-family variants, generated from THEIR OWN reference files (6th element = oracle/ref_shim.py VARIANTS key);
    # for these the stored config is the variant's full config_env + overrides
    "seasonal_short_seed0": ({"season_length_steps": 7, "max_steps": 120}, 0, 1000, False, 8, "seasonal"),
    "seasonal_default_seed1": ({"max_steps": 200}, 1, 1001, False, 20, "seasonal"),
    "plus_eating_seed2": ({"max_steps": 200}, 2, 1002, False, 20, "sparse_rewards_plus_eating"),
    "dense_rewards_seed0": ({"max_steps": 250}, 0, 1000, False, 20, "dense_rewards"),
    "dense_additive_seed4": ({"max_steps": 250}, 4, 1004, False, 20, "dense_rewards_additive"),
    "kickback_seed0": ({"max_steps": 400}, 0, 1000, False, 25, "sparse_rewards_plus_kickback"),
    # fast reproduction + distinctive kickback values: many grandparent rewards (1.25 / 3.0 on top of 0 or 10)
    "kickback_fast_seed5": ({"max_steps": 110, "prey_creation_energy_threshold": 4.5, "predator_creation_energy_threshold": 7.0,
                             "energy_gain_per_step_grass": 0.12, "kickback_reward_predator": 3.0, "kickback_reward_prey": 1.25,
                             "initial_num_grass": 110}, 5, 1005, False, 10, "sparse_rewards_plus_kickback"),
    # drive_conditioned_environment: the base step with extra constant-filled "drive" observation channels
    "drive_default_seed2": ({"max_steps": 120}, 2, 1006, False, 20, "drive_conditioned"),
    "drive_dense_seed3": ({"grid_size": 9, "n_initial_active_predator": 12, "n_initial_active_prey": 20,
                           "initial_num_grass": 25, "predator_obs_range": 5, "prey_obs_range": 7, "max_steps": 150},
                          3, 1007, True, 6, "drive_conditioned"),
    "drive_custom_lists_big_windows_seed4": ({"predator_obs_range": 13, "prey_obs_range": 15, "grid_size": 30,
                                              "n_initial_active_prey": 40, "max_steps": 60,
                                              "predator_drive_channels": ["grass_opportunity", "prey_opportunity"],
                                              "prey_drive_channels": ["predator_danger_pressure"],
                                              "predator_hunger_safe_energy": 4.0, "grass_opportunity_normalizer": 7.5},
                                             4, 1008, False, 10, "drive_conditioned"),
}


def capture(env):
    P, Q, N = env.n_initial_active_predator, env.n_initial_active_prey, env.initial_num_grass
    pred = np.array([env.agent_positions[f"predator_{i}"] for i in range(P)], dtype=np.int32).reshape(P, 2)
    prey = np.array([env.agent_positions[f"prey_{i}"] for i in range(Q)], dtype=np.int32).reshape(Q, 2)
    grass = np.array([env.grass_positions[f"grass_{i}"] for i in range(N)], dtype=np.int32).reshape(N, 2)
    return pred, prey, grass


class _FallbackReached(Exception):
    pass


def _forbid_unseeded_fallback():
    """The reference's spawn fallback draws from the UNSEEDED global np.random (predpreygrass_rllib_env.py:764):
    an episode that reaches it is not reproducible and must not become a golden vector."""
    def boom(*a, **k):
        raise _FallbackReached("episode reaches the unseeded spawn fallback; pick another seed/config")
    np.random.randint = boom


def make_case(name, overrides, seed, action_seed, shuffle, full_every, variant="base", max_calls=1200):
    _forbid_unseeded_fallback()
    env = make_reference_env(overrides, variant)
    if variant != "base":
        overrides = {**{k: v for k, v in reference_default_config(variant).items() if not k.startswith("verbose")},
                     **overrides}
    obs, _ = env.reset(seed=seed)
    pred_xy, prey_xy, grass_xy = capture(env)
    reset_obs = np.concatenate([v.reshape(-1) for v in obs.values()])
    reset_keys = list(obs.keys())
    live = list(obs)
    arng = np.random.default_rng(action_seed)
    rolling = hashlib.sha256()

    act_off, act_type, act_id, act_val = [0], [], [], []
    rec_off, rec_type, rec_id, rec_rew, rec_term, rec_trunc = [0], [], [], [], [], []
    term_all, trunc_all, digests, agents_after = [], [], [], []
    full_calls, obs_off, obs_data, grid_data = [], [0], [], []
    st_off, st_type, st_id, st_x, st_y, st_e, st_cum, st_ate = [0], [], [], [], [], [], [], []
    grass_e_data = []
    t = 0
    fallback_free = True
    while t < max_calls:
        names = list(live)
        if shuffle:
            arng.shuffle(names)
        actions = {a: int(arng.integers(0, 9)) for a in names}
        for a, v in actions.items():
            ty, i = parse_agent(a)
            act_type.append(ty); act_id.append(i); act_val.append(v)
        act_off.append(len(act_type))
        n_before = env._next_predator_idx + env._next_prey_idx
        o, r, te, tr, _ = env.step(actions)
        t += 1
        births = env._next_predator_idx + env._next_prey_idx - n_before
        deaths = sum(1 for k, v in te.items() if k != "__all__" and v)
        for k in o:
            ty, i = parse_agent(k)
            rec_type.append(ty); rec_id.append(i)
            rec_rew.append(float(r[k])); rec_term.append(int(te[k])); rec_trunc.append(int(tr[k]))
        rec_off.append(len(rec_type))
        assert list(o) == list(r) == [k for k in te if k != "__all__"] == [k for k in tr if k != "__all__"]
        term_all.append(int(te["__all__"])); trunc_all.append(int(tr["__all__"]))
        digests.append(np.frombuffer(call_digest(env.grid_world_state, o, r, te, tr), dtype=np.uint8))
        # SURVEY.md Appendix B rolling digest
        rolling.update(env.grid_world_state.tobytes())
        for k, v in o.items():
            rolling.update(k.encode() + v.tobytes())
        for k, v in r.items():
            rolling.update(k.encode() + np.float64(v).tobytes())
        for k, v in te.items():
            rolling.update(k.encode() + bytes([int(v)]))
        agents_after.append(list(env.agents))
        ended = te["__all__"] or tr["__all__"]
        if t <= 3 or t % full_every == 0 or ended or ((births or deaths) and len(full_calls) < 40):
            full_calls.append(t - 1)
            obs_data.append(np.concatenate([v.reshape(-1) for v in o.values()]) if o else np.zeros(0))
            obs_off.append(obs_off[-1] + obs_data[-1].size)
            grid_data.append(env.grid_world_state.copy())
            for a, p in env.agent_positions.items():
                ty, i = parse_agent(a)
                st_type.append(ty); st_id.append(i); st_x.append(int(p[0])); st_y.append(int(p[1]))
                st_e.append(float(env.agent_energies[a])); st_cum.append(float(env.cumulative_rewards[a]))
                st_ate.append(int(a in env.agents_just_ate))
            st_off.append(len(st_type))
            grass_e_data.append(np.array([env.grass_energies[f"grass_{k}"] for k in range(env.initial_num_grass)]))
        live = [a for a in o if not te.get(a, False)]
        if ended:
            break
    out = dict(
        config_json=np.array(json.dumps(overrides)), variant=np.array(variant),
        seed=np.int64(seed), action_seed=np.int64(action_seed), shuffled=np.int8(shuffle),
        pred_xy=pred_xy, prey_xy=prey_xy, grass_xy=grass_xy,
        reset_keys=np.array(json.dumps(reset_keys)), reset_obs_data=reset_obs,
        act_off=np.array(act_off, dtype=np.int32), act_type=np.array(act_type, dtype=np.int8),
        act_id=np.array(act_id, dtype=np.int32), act_val=np.array(act_val, dtype=np.int8),
        rec_off=np.array(rec_off, dtype=np.int32), rec_type=np.array(rec_type, dtype=np.int8),
        rec_id=np.array(rec_id, dtype=np.int32), rec_reward=np.array(rec_rew, dtype=np.float64),
        rec_term=np.array(rec_term, dtype=np.int8), rec_trunc=np.array(rec_trunc, dtype=np.int8),
        term_all=np.array(term_all, dtype=np.int8), trunc_all=np.array(trunc_all, dtype=np.int8),
        digest=np.stack(digests), survey_rolling16=np.array(rolling.hexdigest()[:16]),
        agents_after=np.array(json.dumps(agents_after)),
        full_calls=np.array(full_calls, dtype=np.int32), obs_off=np.array(obs_off, dtype=np.int64),
        obs_data=np.concatenate(obs_data) if obs_data else np.zeros(0),
        grid_data=np.stack(grid_data),
        st_off=np.array(st_off, dtype=np.int32), st_type=np.array(st_type, dtype=np.int8),
        st_id=np.array(st_id, dtype=np.int32), st_x=np.array(st_x, dtype=np.int16), st_y=np.array(st_y, dtype=np.int16),
        st_energy=np.array(st_e, dtype=np.float64), st_cum=np.array(st_cum, dtype=np.float64),
        st_ate=np.array(st_ate, dtype=np.int8), grass_energy=np.stack(grass_e_data),
        final_next_ids=np.array([env._next_predator_idx, env._next_prey_idx], dtype=np.int32),
        final_step=np.int32(env.current_step),
    )
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {t} calls, {len(full_calls)} full, ids {out['final_next_ids'].tolist()}, "
          f"rolling {rolling.hexdigest()[:16]}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        make_case(name, *spec)
