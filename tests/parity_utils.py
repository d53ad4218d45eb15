"""Replay of golden cases / oracle trajectories through the batched tensor API.
Shared by the CPU (wave-emulator) tests and the GPU parity tests."""
from __future__ import annotations

import numpy as np
import torch

from predpreygrass_amd import _abi
from predpreygrass_amd.batched import PREDATOR, PREY, BatchedPredPreyGrass
from tests.golden_io import GoldenCase, call_digest


def obs_of(env: BatchedPredPreyGrass, b, ty, row) -> np.ndarray:
    t = env.obs_pred if ty == PREDATOR else env.obs_prey
    return t[b, row].cpu().numpy().astype(np.float64)


def collect(env: BatchedPredPreyGrass, b, tables=None):
    """(obs, rewards, terminations, truncations) dicts of env b's last call, reference dict order."""
    recs = env.records(b, tables)
    obs, rew, term, trunc = {}, {}, {}, {}
    op = env.obs_pred[b].cpu().numpy()
    oq = env.obs_prey[b].cpu().numpy()
    for name, ty, row, r, te, tr in recs:
        obs[name] = (op if ty == PREDATOR else oq)[row].astype(np.float64)
        rew[name], term[name], trunc[name] = r, te, tr
    i = 0 if tables is None else b
    es = (tables if tables is not None else env.host_tables(b))["env_state"][i]
    fl = int(es[_abi.ENV_FLAGS])
    return recs, obs, rew, term, trunc, bool(fl & _abi.ENVF_TERM_ALL), bool(fl & _abi.ENVF_TRUNC_ALL)


def fill_actions(env: BatchedPredPreyGrass, b, recs, action_dict, strict_order=True):
    """Write env b's action dict into env.actions by the rows of the previous records.
    Returns False if the dict's per-type order differs from row order (not representable)."""
    where = {name: (ty, row) for name, ty, row, _, te, _ in recs if not te}
    a = torch.full((env.S,), _abi.ACTION_NONE, dtype=torch.int8)
    last = {PREDATOR: -1, PREY: -1}
    in_order = True
    for name, act in action_dict.items():
        ty, row = where[name]  # KeyError here == reference KeyError (dead / unknown agent)
        if row < last[ty]:
            in_order = False
        last[ty] = row
        a[row if ty == PREDATOR else env.pred_capacity + row] = int(act)
    env.actions[b].copy_(a)
    return in_order


def replay_golden_cases(make_env, names, defaults, check_grid=True, max_calls=None):
    """Run several golden cases with the SAME config as one batch and compare every call."""
    cases = [GoldenCase(n) for n in names]
    cfg = cases[0].config(defaults)
    for c in cases:
        assert c.config(defaults) == cfg, "cases of one batch must share the config"
    B = len(cases)
    env = make_env(cfg, B)
    env.set_placement(np.stack([c.placement[0] for c in cases]), np.stack([c.placement[1] for c in cases]),
                      np.stack([c.placement[2] for c in cases]))
    recs = []
    for b, c in enumerate(cases):
        r, obs, rew, term, trunc, ta, tra = collect(env, b)
        want = c.reset_obs(cfg)
        assert list(obs) == list(want), (c.name, "reset keys")
        for k in want:
            assert obs[k].tobytes() == want[k].tobytes(), (c.name, "reset obs", k)
        recs.append(r)
    n_calls = max(c.n_calls for c in cases)
    if max_calls:
        n_calls = min(n_calls, max_calls)
    active = [True] * B
    for t in range(n_calls):
        env.actions.fill_(_abi.ACTION_NONE)
        for b, c in enumerate(cases):
            if t >= c.n_calls:
                active[b] = False
                continue
            assert fill_actions(env, b, recs[b], c.actions(t)), (c.name, t, "action order != row order")
        env.step()
        grid = env.export_grid().cpu().numpy() if check_grid else None
        tables = env.host_tables()
        for b, c in enumerate(cases):
            if not active[b]:
                continue
            r, obs, rew, term, trunc, ta, tra = collect(env, b, tables)
            want = c.records(t)
            assert [x[0] for x in r] == [x[0] for x in want], (c.name, t, "dict order", [x[0] for x in r], [x[0] for x in want])
            for (name, _, _, rw, te, tr), (wn, wr, wte, wtr) in zip(r, want):
                assert np.float64(rw).tobytes() == np.float64(wr).tobytes(), (c.name, t, name, "reward", rw, wr)
                assert te == wte and tr == wtr, (c.name, t, name, "flags")
            assert (ta, tra) == c.flags(t), (c.name, t, "__all__")
            status = int(tables["env_state"][b][_abi.ENV_STATUS])
            assert status == 0, (c.name, t, "status", status)
            full = c.full(t, cfg)
            if full is not None:
                fobs, fgrid, fstate, fgrass = full
                for k in fobs:
                    assert obs[k].tobytes() == fobs[k].tobytes(), (c.name, t, "obs", k)
                if check_grid:
                    assert grid[b].tobytes() == fgrid.tobytes(), (c.name, t, "grid")
                ge = tables["grass_energy"][b][: env.n_grass]
                assert ge.tobytes() == fgrass.tobytes(), (c.name, t, "grass energy")
                for (name, ty, row, _, te, _) in r:
                    if te:
                        continue
                    s = row if ty == PREDATOR else env.pred_capacity + row
                    st = fstate[name]
                    xy = int(tables["row_xy"][b][s])
                    assert (xy >> 8, xy & 255) == st["pos"], (c.name, t, name, "pos")
                    assert np.float64(tables["row_energy"][b][s]).tobytes() == np.float64(st["energy"]).tobytes(), (c.name, t, name, "energy")
                    assert np.float64(tables["row_cumrew"][b][s]).tobytes() == np.float64(st["cumulative_reward"]).tobytes(), (c.name, t, name, "cum")
                    assert bool(tables["row_flags"][b][s] & _abi.ROW_ATE) == st["just_ate"], (c.name, t, name, "ate")
            if check_grid:
                te_d = dict(term); te_d["__all__"] = ta
                tr_d = dict(trunc); tr_d["__all__"] = tra
                assert call_digest(grid[b], obs, rew, te_d, tr_d) == c.digest(t), (c.name, t, "digest")
            recs[b] = r
    return env


# ----------------------------------------------------------------------------------------
# random rollouts: device-side Philox actions + auto-reset vs oracle ppo_rollout_random
# ----------------------------------------------------------------------------------------

def compare_env_with_oracle(env: BatchedPredPreyGrass, b, orc, tables, check_obs=True, tag=""):
    """Env b's last call must equal the oracle's last call: dict order, rewards, flags, state, obs."""
    recs = env.records(b, tables)
    orecs, ota, otra = orc.last_records()
    names = [r[0] for r in recs]
    onames = [("predator_%d" if t == 0 else "prey_%d") % i for (t, i, _, _, _) in orecs]
    assert names == onames, (tag, b, "dict order", names, onames)
    for (name, ty, row, rw, te, tr), (_, _, orw, ote, otr) in zip(recs, orecs):
        assert np.float64(rw).tobytes() == np.float64(orw).tobytes(), (tag, b, name, "reward", rw, orw)
        assert te == bool(ote) and tr == bool(otr), (tag, b, name, "flags")
    es = tables["env_state"][b]
    fl = int(es[_abi.ENV_FLAGS])
    assert (bool(fl & _abi.ENVF_TERM_ALL), bool(fl & _abi.ENVF_TRUNC_ALL)) == (ota, otra), (tag, b, "__all__")
    assert int(es[_abi.ENV_STEP]) == orc.current_step, (tag, b, "current_step")
    assert (int(es[_abi.ENV_NEXT_PRED_ID]), int(es[_abi.ENV_NEXT_PREY_ID])) == orc.next_ids, (tag, b, "next ids")
    assert int(es[_abi.ENV_N_PRED_ALIVE]) == orc.current_num_predators, (tag, b)
    assert int(es[_abi.ENV_N_PREY_ALIVE]) == orc.current_num_prey, (tag, b)
    for (name, ty, row, _, te, _) in recs:
        if te:
            continue
        s = row if ty == PREDATOR else env.pred_capacity + row
        st = orc.agent_state(name)
        xy = int(tables["row_xy"][b][s])
        assert (xy >> 8, xy & 255) == st["pos"], (tag, b, name, "pos")
        assert np.float64(tables["row_energy"][b][s]).tobytes() == np.float64(st["energy"]).tobytes(), (tag, b, name, "energy")
        assert np.float64(tables["row_cumrew"][b][s]).tobytes() == np.float64(st["cumulative_reward"]).tobytes(), (tag, b, name, "cum")
    gxy, ge = orc.grass_state()
    assert tables["grass_energy"][b][: env.n_grass].tobytes() == ge.tobytes(), (tag, b, "grass energy")
    gx = tables["grass_xy"][b][: env.n_grass].astype(np.int64)
    assert ((gx >> 8) == gxy[:, 0]).all() and ((gx & 255) == gxy[:, 1]).all(), (tag, b, "grass xy")
    if check_obs:
        o = orc._out
        op = env.obs_pred[b].cpu().numpy()
        oq = env.obs_prey[b].cpu().numpy()
        for k, (name, ty, row, _, _, _) in enumerate(recs):
            r = o.records[k]
            want = np.ctypeslib.as_array(o.obs, shape=(r.obs_offset + r.obs_len,))[r.obs_offset:]
            got = (op if ty == PREDATOR else oq)[row].reshape(-1)
            if got.dtype == np.float64:
                assert got.tobytes() == want.tobytes(), (tag, b, name, "obs")
            else:
                assert (got == want.astype(np.float32)).all(), (tag, b, name, "obs f32")


def rollout_vs_oracle(env: BatchedPredPreyGrass, make_oracle, seed0, n_calls, check_every=1, envs=None, check_grid=False):
    """Step `env` n_calls times with device-side random actions + auto-reset and check it against
    one oracle per env (same Philox contract).  The first call is the reset."""
    B = env.batch_size
    envs = list(range(B)) if envs is None else envs
    oracles = {b: make_oracle() for b in envs}
    env.set_seeds(seed0)
    # mark every env done so that the first auto-reset call performs the reset (episode 0)
    env.env_state.zero_()
    env.env_state[:, _abi.ENV_FLAGS] = _abi.ENVF_DONE
    env.env_state[:, _abi.ENV_EPISODE] = -1
    n_resets = 0
    for t in range(n_calls):
        env.step(random_actions=True, auto_reset=True)
        for b in envs:
            assert oracles[b].rollout_random((seed0 + b) & (2 ** 64 - 1), 1) == 1
        if t % check_every == 0 or t == n_calls - 1:
            tables = env.host_tables()
            grid = env.export_grid().cpu().numpy() if check_grid else None
            for b in envs:
                st = int(tables["env_state"][b][_abi.ENV_STATUS])
                assert st & ~_abi.STATUS_FALLBACK_SPAWN == 0, (t, b, "status", st)
                compare_env_with_oracle(env, b, oracles[b], tables, tag=f"call {t}")
                if check_grid:
                    assert grid[b].tobytes() == oracles[b].grid_world_state.tobytes(), (t, b, "grid")
                n_resets += bool(int(tables["env_state"][b][_abi.ENV_FLAGS]) & _abi.ENVF_WAS_RESET)
    return n_resets
