"""Pins the CPU oracle (oracle/ppg_oracle.c) to the reference's own outputs.

The golden vectors were produced by tests/golden/make_golden.py from
/root/reference/.../base_environment/predpreygrass_rllib_env.py (step() :219-473).
Bit-exact agreement is required on every call: dict ordering, observations,
rewards, flags, the dense grid, env.agents after the sort, and (on sampled
calls) agent positions / energies / cumulative rewards / grass energies.
"""
import numpy as np
import pytest

from oracle.ppg_oracle import DEFAULT_CONFIG, OracleEnv, philox4x32_10
from tests.golden_io import GoldenCase, call_digest, case_names


def f64_equal(a, b):
    return np.float64(a).tobytes() == np.float64(b).tobytes()


@pytest.mark.parametrize("name", case_names())
def test_oracle_replays_golden_case(name):
    case = GoldenCase(name)
    cfg = case.config(DEFAULT_CONFIG)
    env = OracleEnv(cfg)
    obs, info = env.reset_from_placement(*case.placement)
    assert info == {}
    want = case.reset_obs(cfg)
    assert list(obs) == list(want)
    for k in want:
        assert obs[k].tobytes() == want[k].tobytes(), ("reset obs", k)

    for t in range(case.n_calls):
        o, r, te, tr, infos = env.step(case.actions(t))
        assert infos == {}
        assert env.last_fallback_spawns == 0 and env.last_failed_spawns == 0
        recs = case.records(t)
        assert list(o) == [x[0] for x in recs], ("dict order", t)
        assert list(r) == list(o)
        for (k, rew, term, trunc) in recs:
            assert f64_equal(r[k], rew), ("reward", t, k)
            assert te[k] == term and tr[k] == trunc, ("flags", t, k)
        assert (te["__all__"], tr["__all__"]) == case.flags(t)
        assert env.agents == case.agents_after[t], ("agents", t)
        assert call_digest(env.grid_world_state, o, r, te, tr) == case.digest(t), ("digest", t)
        full = case.full(t, cfg)
        if full is not None:
            fobs, grid, state, grass_e = full
            assert env.grid_world_state.tobytes() == grid.tobytes()
            for k in fobs:
                assert o[k].tobytes() == fobs[k].tobytes(), ("obs", t, k)
            for k, s in state.items():
                got = env.agent_state(k)
                assert got["pos"] == s["pos"] and got["just_ate"] == s["just_ate"]
                assert f64_equal(got["energy"], s["energy"]), ("energy", t, k)
                assert f64_equal(got["cumulative_reward"], s["cumulative_reward"]), ("cum", t, k)
            assert env.grass_state()[1].tobytes() == grass_e.tobytes()
    assert env.current_step == int(case.z["final_step"])
    assert list(env.next_ids) == case.z["final_next_ids"].tolist()


def test_survey_known_answers():
    """rolling digests of the reference under the same protocol."""
    assert str(GoldenCase("c1_seed0").z["survey_rolling16"]) == "02b8208df708d78f"
    assert str(GoldenCase("default_seed0").z["survey_rolling16"]) == "53b9397117d97813"
    assert GoldenCase("c1_seed0").n_calls == 34 and GoldenCase("default_seed0").n_calls == 1001


def test_obs_clip_known_answers():
    """SURVEY.md Appendix B `_obs_clip` table (predpreygrass_rllib_env.py:528-539) via the obs mask."""
    env = OracleEnv({"n_initial_active_predator": 1, "n_initial_active_prey": 1, "initial_num_grass": 1})
    cases = [((0, 0), "predator_0", 7, (3, 7, 3, 7)), ((24, 24), "predator_0", 7, (0, 4, 0, 4)),
             ((0, 12), "prey_0", 9, (4, 9, 0, 9)), ((3, 3), "predator_0", 7, (0, 7, 0, 7)),
             ((1, 23), "prey_0", 9, (3, 9, 0, 6))]
    for (x, y), who, R, (xolo, xohi, yolo, yohi) in cases:
        pred = [(x, y)] if who.startswith("pred") else [(10, 10)]
        prey = [(x, y)] if who.startswith("prey") else [(11, 11)]
        obs, _ = env.reset_from_placement(pred, prey, [(5, 5)])
        mask = obs[who][0]
        want = np.ones((R, R))
        want[xolo:xohi, yolo:yohi] = 0
        assert (mask == want).all(), (x, y)


def test_action_for_dead_agent_raises_keyerror():
    """E12: predpreygrass_rllib_env.py:246/249."""
    env = OracleEnv({"n_initial_active_predator": 1, "n_initial_active_prey": 1, "initial_num_grass": 1})
    env.reset_from_placement([(1, 1)], [(5, 5)], [(9, 9)])
    with pytest.raises(KeyError):
        env.step({"prey_7": 4})


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    assert philox4x32_10([0, 0, 0, 0], [0, 0]).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    ff = 0xFFFFFFFF
    assert philox4x32_10([ff, ff, ff, ff], [ff, ff]).tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_reset_places_unique_cells_and_rollout_runs():
    env = OracleEnv({})
    obs, _ = env.reset_philox(seed=123, episode=0)
    assert len(obs) == 14
    cells = set(env.agent_positions.values())
    gxy, ge = env.grass_state()
    cells |= {tuple(p) for p in gxy.tolist()}
    assert len(cells) == 6 + 8 + 100 and (ge == 2.0).all()
    env2 = OracleEnv({})
    assert env2.rollout_random(seed=7, n_calls=400) == 400
