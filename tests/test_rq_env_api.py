"""The red_queen-shaped dict API (predpreygrass_amd.red_queen.PredPreyGrass) on the wave emulator: golden episodes
replayed from the SEED ALONE (placement and reproduction uniforms come out of the same generators as in the
reference), attributes, snapshot/restore, error behaviour."""
import numpy as np
import pytest

from predpreygrass_amd.red_queen import PredPreyGrass, config_env_base
from tests.emu_backend import library
from tests.golden_io_rq import RQGoldenCase, call_digest, case_names


def make(cfg, **kw):
    return PredPreyGrass(cfg, _library=library(), **kw)


@pytest.mark.parametrize("name", case_names())
def test_dict_api_replays_reference_from_seed(name):
    case = RQGoldenCase(name)
    env = make(case.config)
    obs, info = env.reset(seed=int(case.z["seed"]))
    assert info == {}
    want = case.reset_obs()
    assert list(obs) == list(want) == env.agents
    for k in want:
        assert obs[k].dtype == np.float32 and obs[k].tobytes() == want[k].tobytes()
    for t in range(min(case.n_calls, 100)):
        o, r, te, tr, infos = env.step(case.actions(t))
        assert infos == {}
        recs = case.records(t)
        assert list(o) == [x[0] for x in recs] == list(r), (name, t)
        assert list(te) == [x[0] for x in recs] + ["__all__"] and list(tr) == list(te)
        for k, rew, term, trunc in recs:
            assert np.float64(r[k]).tobytes() == np.float64(rew).tobytes() and te[k] is term and tr[k] is trunc
        assert (te["__all__"], tr["__all__"]) == case.flags(t)
        assert env.agents == case.agents_after[t], (name, t)
        assert call_digest(env.grid_world_state, o, r, te, tr) == case.digest(t), (name, t)
        full = case.full(t)
        if full is not None:
            _, _, state, grass_e, next_idx = full
            pos, en, lr = env.agent_positions, env.agent_energies, env.agent_last_reproduction
            assert list(pos) == list(state), (name, t, "agent_positions insertion order")
            for k, s in state.items():
                assert pos[k] == s["pos"] and en[k] == s["energy"] and lr[k] == s["last_reproduction"]
                assert env.cumulative_rewards[k] == s["cumulative_reward"] and (k in env.agents_just_ate) == s["just_ate"]
            assert list(env.grass_energies.values()) == grass_e.tolist()
            assert tuple(env._next_idx[(s, ty)] for s in ("predator", "prey") for ty in (1, 2)) == next_idx


def test_spaces_and_attributes():
    env = make(config_env_base)
    assert env.action_spaces["type_1_prey_3"].n == 9 and env.action_spaces["type_2_prey_3"].n == 25
    assert env.observation_spaces["type_1_predator_0"].shape == (4, 7, 7)
    assert env.observation_spaces["type_2_prey_0"].shape == (4, 9, 9)
    assert len(env.possible_agents) == 2000 + 0 + 1600 + 1600
    obs, _ = env.reset(seed=1)
    assert len(obs) == 32 and env.active_num_predators == 12 and env.active_num_prey == 20
    tot = env.get_total_energy_by_type()
    assert tot["predator"] == 12 * 6.0 and tot["type_2_prey"] == 10 * 3.0 and tot["grass"] == 200.0


def test_missing_or_bad_actions_raise_like_the_reference():
    env = make(config_env_base)
    obs, _ = env.reset(seed=2)
    acts = {a: 0 for a in obs}
    with pytest.raises(KeyError):
        env.step({a: 9 for a in obs})          # 9 is outside a type-1 agent's 3x3 action space
    first = next(iter(acts))
    with pytest.raises(KeyError):
        env.step({a: v for a, v in acts.items() if a != first})   # a live agent without an action
    acts["type_1_prey_999"] = 3                  # an agent that does not exist is ignored (RQ:467,521)
    env.step(acts)
    with pytest.raises(ValueError):
        PredPreyGrass(None, _library=library())


def test_snapshot_restore_roundtrip():
    case = RQGoldenCase("rq_mixed_types_seed7")
    env = make(case.config)
    env.reset(seed=int(case.z["seed"]))
    for t in range(20):
        env.step(case.actions(t))
    snap = env.get_state_snapshot()
    a = [env.step(case.actions(t)) for t in range(20, 30)]
    env.restore_state_snapshot(snap)
    b = [env.step(case.actions(t)) for t in range(20, 30)]
    for x, y in zip(a, b):
        assert list(x[0]) == list(y[0]) and x[1] == y[1] and x[2] == y[2]
        for k in x[0]:
            assert x[0][k].tobytes() == y[0][k].tobytes()
