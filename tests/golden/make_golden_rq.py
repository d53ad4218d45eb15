#!/usr/bin/env python3
"""Generate the second-generation golden vectors (tests/golden/rq/*.npz) FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the produced files are data (inputs + expected
outputs) and are what travels to the GPU box.

    python tests/golden/make_golden_rq.py

The reference env /root/reference/predpreygrass/non_evolutionary/red_queen/predpreygrass_rllib_env.py (imported
through oracle/ref_shim.py) is reset with a seed; its initial placement is captured (it depends on CPython's set
order and is treated as captured input); `env.rng` -- numpy PCG64 seeded by reset(), the source of the
reproduction-chance and mutation uniforms (RQ:701,708,786,793) -- is wrapped so that every value it returns is
recorded; the env is then driven with the live-agent protocol (actions only for agents that got an observation
last call and were not terminated), one `default_rng(action_seed).integers(n_actions)` per agent in dict order.
A case that reaches the set-order dependent spawn fallback (RQ:396-401) is rejected.

Cases with variant "walls_occlusion" (tests/golden/wo/*.npz) come from
/root/reference/predpreygrass/non_evolutionary/walls_occlusion/predpreygrass_rllib_env.py ("WO": the same env plus static
walls, line-of-sight rules and per-agent infos).  Extra fields: wall_xy (the episode's walls, sorted), info_off[T+1] /
info_pool / info_id / info_reason (infos dict: 0 = {"los_rejected": 0}, 1 wall, 2 occupied, 3 corner_cut, 4 los), and
extra_off[T+1] / extra_pool / extra_id: agents that appear in the scalar dicts only (named in action_dict but gone; reward
0.0, False, False).  That env builds its scalar dicts from a Python set (WO:376-388), so their ORDER is arbitrary in the
reference itself: rec_* keep the observation-dict order and the digest hashes the scalar dicts in sorted key order.

Stored per case:
  config_json                      the complete config dict
  pred_xy/prey_xy/grass_xy         captured placement (predators / prey in creation order, grass in id order)
  act_off[T+1], act_pool/act_id/act_val      action dicts, flattened, dict order (pool: 0 type_1_predator,
                                   1 type_2_predator, 2 type_1_prey, 3 type_2_prey)
  uniforms[], uni_off[T+1]         the recorded uniform stream and how much of it each call consumed
  rec_off[T+1], rec_pool/rec_id/rec_reward/rec_term/rec_trunc, term_all[T], trunc_all[T]   returned dicts
  digest[T,32]                     sha256 per call (`call_digest`)
  agents_after (json)              env.agents after each call
  full_calls[], obs_off[], obs_data (float32), grid_data (float32 (4,G,G)), st_* agent state incl. age and
                                   last reproduction step, grass_energy, next_idx   for the calls in full_calls
  reset_keys (json), reset_obs_data
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.ref_shim import make_reference_env, reference_config, reference_default_config  # noqa: E402

POOLS = ("type_1_predator", "type_2_predator", "type_1_prey", "type_2_prey")
OUT_DIR = os.path.join(HERE, "rq")


def parse_agent(name):
    kind, idx = name.rsplit("_", 1)
    return POOLS.index(kind), int(idx)


def call_digest(grid, obs, rew, term, trunc, sort_scalars=False) -> bytes:
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(grid, dtype=np.float32).tobytes())
    for k, v in obs.items():
        h.update(k.encode() + np.ascontiguousarray(v, dtype=np.float32).tobytes())
    if sort_scalars:
        rew, term, trunc = ({k: d[k] for k in sorted(d)} for d in (rew, term, trunc))
    for k, v in rew.items():
        h.update(k.encode() + np.float64(v).tobytes())
    for k, v in term.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    for k, v in trunc.items():
        h.update(k.encode() + bytes([int(bool(v))]))
    return h.digest()


class RecordingRng:
    """Stands in for env.rng after reset(): records what random() returns; the fallback's integers() is refused."""

    def __init__(self, inner):
        self.inner = inner
        self.values = []

    def random(self):
        v = self.inner.random()
        self.values.append(float(v))
        return v

    def integers(self, *a, **k):
        raise RuntimeError("golden case reached the set-order dependent spawn fallback (RQ:396-401); change its parameters")

    def choice(self, *a, **k):
        raise RuntimeError("rng.choice is only expected inside reset() (WO:246,260)")


def _train_v1():
    return reference_config("predpreygrass.non_evolutionary.red_queen.config.config_env_train_v1_0")


def _eval():
    return reference_config("predpreygrass.non_evolutionary.red_queen.config.config_env_eval")


_TYPED = {
    "reward_prey_eat_grass": {"type_1_prey": 0.5, "type_2_prey": 1.5},
    "penalty_prey_caught": {"type_1_prey": -1.0, "type_2_prey": -2.0},
    "reward_predator_catch_prey": {"type_1_predator": 2.0, "type_2_predator": 3.0},
    "reward_prey_step": 0.125,
    "reward_predator_step": {"type_1_predator": -0.25, "type_2_predator": 0.0625},
    "reproduction_reward_predator": {"type_1_predator": 10.0, "type_2_predator": 7.5},
    "reproduction_reward_prey": {"type_1_prey": 10.0, "type_2_prey": 4.25},
}

CASES = {
    # name: (config builder, reset seed, action seed, action-dict mode, full-every, max calls)
    "rq_base_seed3": (lambda: {}, 3, 0, "live", 40, 1200),
    "rq_train_v1_seed1": (_train_v1, 1, 5, "live", 20, 1200),
    "rq_eval_seed0": (_eval, 0, 11, "live", 25, 400),
    "rq_mixed_types_seed7": (lambda: dict(
        _TYPED, grid_size=14, max_steps=160, initial_num_grass=60, predator_obs_range=5, prey_obs_range=7,
        n_initial_active_type_1_predator=5, n_initial_active_type_2_predator=5,
        n_initial_active_type_1_prey=9, n_initial_active_type_2_prey=9,
        n_possible_type_1_predators=40, n_possible_type_2_predators=40,
        n_possible_type_1_prey=60, n_possible_type_2_prey=60,
        mutation_rate_predator=0.4, mutation_rate_prey=0.4, move_energy_cost_factor=0.04,
        reproduction_cooldown_steps=3, reproduction_chance_predator=0.7, reproduction_chance_prey=0.8,
        energy_gain_per_step_grass=0.25, energy_loss_per_step_predator=0.1, energy_loss_per_step_prey=0.03,
        initial_energy_predator=7.0, predator_creation_energy_threshold=9.0, prey_creation_energy_threshold=5.0,
        max_energy_gain_per_grass=1.25, max_energy_gain_per_prey=3.5, max_energy_predator=11.0, max_energy_prey=6.5,
        energy_transfer_efficiency=0.85, reproduction_energy_efficiency=0.8), 7, 9, "live", 8, 400),
    "rq_pool_exhaust_seed2": (lambda: dict(
        _TYPED, grid_size=12, max_steps=140, initial_num_grass=50, predator_obs_range=7, prey_obs_range=7,
        n_initial_active_type_1_predator=4, n_initial_active_type_2_predator=3,
        n_initial_active_type_1_prey=8, n_initial_active_type_2_prey=6,
        n_possible_type_1_predators=6, n_possible_type_2_predators=4,
        n_possible_type_1_prey=14, n_possible_type_2_prey=9,
        mutation_rate_predator=0.3, mutation_rate_prey=0.3, move_energy_cost_factor=0.02,
        reproduction_cooldown_steps=2, reproduction_chance_predator=0.9, reproduction_chance_prey=0.9,
        energy_gain_per_step_grass=0.3, energy_loss_per_step_predator=0.08, energy_loss_per_step_prey=0.02,
        predator_creation_energy_threshold=7.0, prey_creation_energy_threshold=4.5,
        max_energy_gain_per_grass=2.0, max_energy_gain_per_prey=4.0, max_energy_predator=15.0, max_energy_prey=9.0,
        energy_transfer_efficiency=0.9, reproduction_energy_efficiency=0.75), 2, 21, "live", 10, 400),
    "rq_shuffled_seed5": (lambda: dict(
        _TYPED, grid_size=10, max_steps=90, initial_num_grass=35, predator_obs_range=5, prey_obs_range=5,
        n_initial_active_type_1_predator=6, n_initial_active_type_2_predator=6,
        n_initial_active_type_1_prey=10, n_initial_active_type_2_prey=10,
        n_possible_type_1_predators=30, n_possible_type_2_predators=30,
        n_possible_type_1_prey=50, n_possible_type_2_prey=50,
        mutation_rate_predator=0.2, mutation_rate_prey=0.2, move_energy_cost_factor=0.03,
        reproduction_cooldown_steps=4, reproduction_chance_predator=0.85, reproduction_chance_prey=0.85,
        energy_gain_per_step_grass=0.2, initial_energy_predator=6.0, predator_creation_energy_threshold=8.0,
        prey_creation_energy_threshold=5.0, energy_loss_per_step_predator=0.2, energy_loss_per_step_prey=0.1,
        max_energy_gain_per_grass=1.5, max_energy_gain_per_prey=5.0, max_energy_predator=20.0, max_energy_prey=14.0,
        energy_transfer_efficiency=0.9, reproduction_energy_efficiency=0.9), 5, 33, "shuffled", 6, 300),
    "rq_trunc_seed4": (lambda: dict(max_steps=25, move_energy_cost_factor=0.01), 4, 3, "live", 5, 40),
    # ---- walls_occlusion env ----
    "wo_zigzag_seed1": (lambda: reference_config("predpreygrass.non_evolutionary.walls_occlusion.config.config_env_zigzag_walls"),
                        1, 0, "live", 10, 300, "walls_occlusion"),
    "wo_perimeter_seed2": (lambda: reference_config(
        "predpreygrass.non_evolutionary.walls_occlusion.config.config_env_perimeter_four_gaps_walls"), 2, 4, "live", 12, 200,
        "walls_occlusion"),
    "wo_base_random_walls_seed3": (lambda: {}, 3, 1, "live", 25, 150, "walls_occlusion"),
    "wo_los_two_types_seed5": (lambda: dict(
        _TYPED, num_walls=60, respect_los_for_movement=True, mask_observation_with_visibility=True,
        include_visibility_channel=True, type_2_action_range=5, grid_size=14, initial_num_grass=40, max_steps=80,
        n_initial_active_type_2_predator=4, n_possible_type_2_predators=40, energy_gain_per_step_grass=0.3,
        predator_creation_energy_threshold=8.0, prey_creation_energy_threshold=4.5), 5, 2, "live", 8, 120, "walls_occlusion"),
    "wo_mask_only_shuffled_seed6": (lambda: dict(
        _TYPED, num_walls=45, respect_los_for_movement=True, mask_observation_with_visibility=True,
        include_visibility_channel=False, predator_obs_range=6, prey_obs_range=8, type_2_action_range=5, grid_size=13,
        initial_num_grass=35, max_steps=70, n_initial_active_type_2_predator=3, n_possible_type_2_predators=30,
        energy_gain_per_step_grass=0.3, prey_creation_energy_threshold=4.5), 6, 3, "shuffled", 7, 100, "walls_occlusion"),
}
REASON_CODE = {None: 0, "wall": 1, "occupied": 2, "corner_cut": 3, "los": 4}


def capture(env):
    pred = np.array([env.agent_positions[a] for a in env.agents if "predator" in a], dtype=np.int32).reshape(-1, 2)
    prey = np.array([env.agent_positions[a] for a in env.agents if "prey" in a], dtype=np.int32).reshape(-1, 2)
    grass = np.array([env.grass_positions[g] for g in env.grass_agents], dtype=np.int32).reshape(-1, 2)
    return pred, prey, grass


def make_case(name, build_cfg, seed, action_seed, mode, full_every, max_calls, variant="red_queen"):
    wo = variant == "walls_occlusion"
    cfg = reference_default_config(variant)
    cfg.update(build_cfg())
    env = make_reference_env(cfg, variant)
    cfg = dict(env.config)
    obs, _ = env.reset(seed=seed)
    env.rng = RecordingRng(env.rng)
    pred_xy, prey_xy, grass_xy = capture(env)
    wall_xy = np.array(sorted(env.wall_positions), dtype=np.int32).reshape(-1, 2) if wo else np.zeros((0, 2), dtype=np.int32)
    info_off, info_pool, info_id, info_reason = [0], [], [], []
    extra_off, extra_pool, extra_id = [0], [], []
    reset_keys = list(obs)
    reset_obs = np.concatenate([obs[k].reshape(-1) for k in reset_keys]).astype(np.float32)
    arng = np.random.default_rng(action_seed)

    act_off, act_pool, act_id, act_val = [0], [], [], []
    uni_off = [0]
    rec_off, rec_pool, rec_id, rec_rew, rec_term, rec_trunc = [0], [], [], [], [], []
    term_all, trunc_all, digests, agents_after = [], [], [], []
    full_calls, obs_off, obs_data, grid_data = [], [0], [], []
    st_off, st_pool, st_id, st_x, st_y, st_e, st_cum, st_ate, st_age, st_lr = [0], [], [], [], [], [], [], [], [], []
    grass_energy, next_idx = [], []
    stats = dict(births=0, mutations=0, ate_then_caught=0)

    live = list(obs)
    dead_pool = []
    for t in range(max_calls):
        names = list(live)
        if mode == "shuffled":
            # any order, sometimes with extra actions for agents that are already dead (the red_queen env skips
            # those, RQ:467,521).  Every live agent must act: the reference itself raises KeyError at RQ:279
            # (per-step analytics) when one is missing, so strict subsets are not valid inputs.
            arng.shuffle(names)
            if t % 4 == 2 and dead_pool:
                names.insert(int(arng.integers(len(names) + 1)), dead_pool[int(arng.integers(len(dead_pool)))])
        act = {a: int(arng.integers(env.action_spaces[a].n)) for a in names}
        for a, v in act.items():
            p, i = parse_agent(a)
            act_pool.append(p); act_id.append(i); act_val.append(v)
        act_off.append(len(act_val))
        before = dict(env._next_idx)
        o, r, te, tr, infos = env.step(act)
        uni_off.append(len(env.rng.values))
        if wo:
            for a in sorted(infos):
                p, i = parse_agent(a)
                assert set(infos[a]) <= {"los_rejected", "move_blocked_reason"}
                code = REASON_CODE[infos[a].get("move_blocked_reason")]
                assert infos[a]["los_rejected"] == int(code == 4)
                info_pool.append(p); info_id.append(i); info_reason.append(code)
            for a in sorted(r):
                if a not in o:
                    assert r[a] == 0.0 and te[a] is False and tr[a] is False
                    p, i = parse_agent(a)
                    extra_pool.append(p); extra_id.append(i)
            assert set(r) == set(te) - {"__all__"} == set(tr) - {"__all__"}
        else:
            assert infos == {}
        info_off.append(len(info_id)); extra_off.append(len(extra_id))
        for a in o:
            p, i = parse_agent(a)
            rec_pool.append(p); rec_id.append(i); rec_rew.append(float(r[a]))
            rec_term.append(bool(te[a])); rec_trunc.append(bool(tr[a]))
            if te[a] and a in env.agents_just_ate and "prey" in a:
                stats["ate_then_caught"] += 1
        rec_off.append(len(rec_id))
        term_all.append(bool(te["__all__"])); trunc_all.append(bool(tr["__all__"]))
        digests.append(np.frombuffer(call_digest(env.grid_world_state, o, r, te, tr, sort_scalars=wo), dtype=np.uint8))
        agents_after.append(list(env.agents))
        stats["births"] += sum(env._next_idx[k] - before[k] for k in before)
        last = te["__all__"] or tr["__all__"] or t == max_calls - 1
        if t % full_every == 0 or last or t < 3:
            full_calls.append(t)
            for a in o:
                obs_data.append(o[a].reshape(-1).astype(np.float32))
            obs_off.append(obs_off[-1] + sum(o[a].size for a in o))
            grid_data.append(env.grid_world_state.copy())
            for a in env.agent_positions:
                p, i = parse_agent(a)
                st_pool.append(p); st_id.append(i)
                st_x.append(int(env.agent_positions[a][0])); st_y.append(int(env.agent_positions[a][1]))
                st_e.append(float(env.agent_energies[a])); st_cum.append(float(env.cumulative_rewards.get(a, 0.0)))
                st_ate.append(a in env.agents_just_ate)
                st_age.append(int(env.agent_ages[a])); st_lr.append(int(env.agent_last_reproduction[a]))
            st_off.append(len(st_id))
            grass_energy.append(np.array([env.grass_energies[g] for g in env.grass_agents], dtype=np.float64))
            next_idx.append([env._next_idx[(s, ty)] for s in ("predator", "prey") for ty in (1, 2)])
        for a in o:
            if te[a]:
                dead_pool.append(a)
        live = [a for a in o if not te[a]]
        if te["__all__"] or tr["__all__"]:
            break
    for uid, st in env.unique_agent_stats.items():
        stats["mutations"] += bool(st.get("mutated"))
    out_dir = os.path.join(HERE, "wo") if wo else OUT_DIR
    os.makedirs(out_dir, exist_ok=True)
    stats["los_rejected"] = int(sum(1 for v in info_reason if v == 4))
    stats["blocked"] = {k: int(sum(1 for v in info_reason if v == c)) for k, c in REASON_CODE.items() if c}
    np.savez_compressed(
        os.path.join(out_dir, name + ".npz"),
        variant=variant, wall_xy=wall_xy,
        info_off=np.array(info_off, dtype=np.int64), info_pool=np.array(info_pool, dtype=np.int8),
        info_id=np.array(info_id, dtype=np.int32), info_reason=np.array(info_reason, dtype=np.int8),
        extra_off=np.array(extra_off, dtype=np.int64), extra_pool=np.array(extra_pool, dtype=np.int8),
        extra_id=np.array(extra_id, dtype=np.int32),
        config_json=json.dumps(cfg), seed=seed, action_seed=action_seed,
        pred_xy=pred_xy, prey_xy=prey_xy, grass_xy=grass_xy,
        act_off=np.array(act_off, dtype=np.int64), act_pool=np.array(act_pool, dtype=np.int8),
        act_id=np.array(act_id, dtype=np.int32), act_val=np.array(act_val, dtype=np.int8),
        uniforms=np.array(env.rng.values, dtype=np.float64), uni_off=np.array(uni_off, dtype=np.int64),
        rec_off=np.array(rec_off, dtype=np.int64), rec_pool=np.array(rec_pool, dtype=np.int8),
        rec_id=np.array(rec_id, dtype=np.int32), rec_reward=np.array(rec_rew, dtype=np.float64),
        rec_term=np.array(rec_term, dtype=bool), rec_trunc=np.array(rec_trunc, dtype=bool),
        term_all=np.array(term_all, dtype=bool), trunc_all=np.array(trunc_all, dtype=bool),
        digest=np.stack(digests), agents_after=json.dumps(agents_after),
        full_calls=np.array(full_calls, dtype=np.int64), obs_off=np.array(obs_off, dtype=np.int64),
        obs_data=np.concatenate(obs_data) if obs_data else np.zeros(0, dtype=np.float32),
        grid_data=np.stack(grid_data).astype(np.float32),
        st_off=np.array(st_off, dtype=np.int64), st_pool=np.array(st_pool, dtype=np.int8),
        st_id=np.array(st_id, dtype=np.int32), st_x=np.array(st_x, dtype=np.int16), st_y=np.array(st_y, dtype=np.int16),
        st_energy=np.array(st_e, dtype=np.float64), st_cum=np.array(st_cum, dtype=np.float64),
        st_ate=np.array(st_ate, dtype=bool), st_age=np.array(st_age, dtype=np.int32),
        st_last_repro=np.array(st_lr, dtype=np.int32),
        grass_energy=np.stack(grass_energy), next_idx=np.array(next_idx, dtype=np.int32),
        reset_keys=json.dumps(reset_keys), reset_obs_data=reset_obs,
    )
    size = os.path.getsize(os.path.join(out_dir, name + ".npz"))
    print(f"{name}: {len(term_all)} calls, {len(env.rng.values)} uniforms, {stats}, next_idx={dict(env._next_idx)}, "
          f"{size / 1024:.0f} KiB")


def main():
    only = sys.argv[1:]
    for name, args in CASES.items():
        if only and name not in only:
            continue
        make_case(name, *args)


if __name__ == "__main__":
    main()
