"""Replay of second-generation golden cases / oracle trajectories through `BatchedRedQueen`.
Shared by the CPU (wave-emulator) tests and the GPU parity tests."""
from __future__ import annotations

import numpy as np
import torch

from predpreygrass_amd import _abi
from predpreygrass_amd.red_queen import BatchedRedQueen, agent_name, split_row_id
from tests.golden_io_rq import RQGoldenCase, call_digest


def collect(env: BatchedRedQueen, b, tables=None):
    recs = env.records(b, tables)
    obs, rew, term, trunc = {}, {}, {}, {}
    op = env.obs_pred[b].cpu().numpy()
    oq = env.obs_prey[b].cpu().numpy()
    for name, sp, row, r, te, tr in recs:
        obs[name] = (oq if sp else op)[row]
        rew[name], term[name], trunc[name] = r, te, tr
    i = 0 if tables is None else b
    es = (tables if tables is not None else env.host_tables(b))["env_state"][i]
    fl = int(es[_abi.ENV_FLAGS])
    return recs, obs, rew, term, trunc, bool(fl & _abi.ENVF_TERM_ALL), bool(fl & _abi.ENVF_TRUNC_ALL)


def infos_of(env: BatchedRedQueen, b, recs, tables):
    """infos dict of the walls env (WO:766-778) from row_info."""
    out = {}
    for name, sp, row, *_ in recs:
        code = int(tables["row_info"][b][env.pred_capacity * sp + row])
        if code:
            d = {"los_rejected": int(code - 1 == 4)}
            if code > 1:
                d["move_blocked_reason"] = _abi.MOVE_REASONS[code - 1]
            out[name] = d
    return out


def fill_actions(env: BatchedRedQueen, b, recs, action_dict, rank):
    """Write env b's action dict into env.actions (rows of the previous records) and the per-species action order
    into `rank`.  Actions naming agents that are no longer alive are dropped, like the reference does (RQ:467,521).
    Returns True if the dict's per-species order equals row order."""
    where = {name: (sp, row) for name, sp, row, _, te, _ in recs if not te}
    a = torch.full((env.S,), _abi.ACTION_NONE, dtype=torch.int8)
    rk = torch.zeros((env.S,), dtype=torch.uint8)
    last = [-1, -1]
    count = [0, 0]
    in_order = True
    for name, act in action_dict.items():
        if name not in where:
            continue
        sp, row = where[name]
        if row < last[sp]:
            in_order = False
        last[sp] = row
        s = env.pred_capacity * sp + row
        a[s] = int(act)
        rk[s] = count[sp]
        count[sp] += 1
    env.actions[b].copy_(a)
    rank[b].copy_(rk)
    return in_order


def replay_golden_case(make_env, name, max_calls=None, extra_uniforms=2):
    """One golden case through BatchedRedQueen (batch 1), every call compared with what the reference returned."""
    c = RQGoldenCase(name)
    cfg = c.config
    env = make_env(cfg, 1, **({"walls": True} if c.walls else {}))
    if c.walls:
        env.set_walls(c.wall_xy)
    env.set_placement(*[np.asarray(a)[None] for a in c.placement])
    recs, obs, rew, term, trunc, ta, tra = collect(env, 0)
    want = c.reset_obs()
    assert list(obs) == list(want), (name, "reset keys", list(obs), list(want))
    for k in want:
        assert obs[k].dtype == np.float32 and obs[k].tobytes() == want[k].tobytes(), (name, "reset obs", k)
    n_calls = c.n_calls if not max_calls else min(c.n_calls, max_calls)
    rank = torch.zeros((1, env.S), dtype=torch.uint8, device=env.device)
    n_ordered = 0
    for t in range(n_calls):
        in_order = fill_actions(env, 0, recs, c.actions(t), rank)
        u, n_used = c.uniforms(t, extra=extra_uniforms)
        ut = torch.zeros((1, max(len(u), 1)), dtype=torch.float64, device=env.device)
        ut[0, : len(u)] = torch.from_numpy(np.ascontiguousarray(u))
        env.step(uniforms=ut, act_rank=None if in_order else rank)
        n_ordered += not in_order
        tables = env.host_tables()
        grid = env.export_grid().cpu().numpy()[0].astype(np.float32)
        recs, obs, rew, term, trunc, ta, tra = collect(env, 0, tables)
        es = tables["env_state"][0]
        wantr = c.records(t)
        assert [x[0] for x in recs] == [x[0] for x in wantr], (name, t, "dict order", [x[0] for x in recs], [x[0] for x in wantr])
        for (nm, _, _, rw, te, tr), (wn, wr, wte, wtr) in zip(recs, wantr):
            assert np.float64(rw).tobytes() == np.float64(wr).tobytes(), (name, t, nm, "reward", rw, wr)
            assert te == wte and tr == wtr, (name, t, nm, "flags")
        assert (ta, tra) == c.flags(t), (name, t, "__all__")
        assert int(es[_abi.ENV_STATUS]) == 0, (name, t, "status", int(es[_abi.ENV_STATUS]))
        if not tra:
            assert int(es[_abi.ENV_DRAWS]) == n_used, (name, t, "draws", int(es[_abi.ENV_DRAWS]), n_used)
        te_d = dict(term); te_d["__all__"] = ta
        tr_d = dict(trunc); tr_d["__all__"] = tra
        if c.walls:   # scalar dicts also carry the agents named in the action dict that are gone (WO:373-388)
            assert infos_of(env, 0, recs, tables) == c.infos(t), (name, t, "infos", infos_of(env, 0, recs, tables), c.infos(t))
            if not tra:
                for a in c.actions(t):
                    if a not in rew:
                        rew[a], te_d[a], tr_d[a] = 0.0, False, False
            assert sorted(a for a in rew if a not in obs) == sorted(c.extras(t)), (name, t, "extras")
        assert call_digest(grid, obs, rew, te_d, tr_d, sort_scalars=c.walls) == c.digest(t), (name, t, "digest")
        full = c.full(t)
        if full is not None:
            fobs, fgrid, fstate, fgrass, next_idx = full
            for k in fobs:
                assert obs[k].tobytes() == fobs[k].tobytes(), (name, t, "obs", k)
            assert grid.tobytes() == fgrid.tobytes(), (name, t, "grid")
            assert tables["grass_energy"][0][: env.n_grass].tobytes() == fgrass.tobytes(), (name, t, "grass energy")
            got_next = (int(es[_abi.ENV_NEXT_PRED_ID]), int(es[_abi.ENV_NEXT_PRED_ID_T2]), int(es[_abi.ENV_NEXT_PREY_ID]),
                        int(es[_abi.ENV_NEXT_PREY_ID_T2]))
            assert got_next == next_idx, (name, t, "next_idx", got_next, next_idx)
            for (nm, sp, row, _, te, _) in recs:
                if te:
                    continue
                s = env.pred_capacity * sp + row
                st = fstate[nm]
                xy = int(tables["row_xy"][0][s])
                assert (xy >> 8, xy & 255) == st["pos"], (name, t, nm, "pos")
                assert np.float64(tables["row_energy"][0][s]).tobytes() == np.float64(st["energy"]).tobytes(), (name, t, nm, "energy")
                assert np.float64(tables["row_cumrew"][0][s]).tobytes() == np.float64(st["cumulative_reward"]).tobytes(), (name, t, nm, "cum")
                assert bool(tables["row_flags"][0][s] & _abi.ROW_ATE) == st["just_ate"], (name, t, nm, "ate")
                assert int(tables["row_lastrep"][0][s]) == st["last_reproduction"], (name, t, nm, "last_reproduction")
    return env, n_ordered


def compare_env_with_oracle(env: BatchedRedQueen, b, orc, tables, tag=""):
    """Env b's last call must equal the oracle's last call: dict order, rewards, flags, state, observations."""
    recs = env.records(b, tables)
    orecs, ota, otra = orc.last_records()
    names = [r[0] for r in recs]
    onames = [agent_name(p, i) for (p, i, _, _, _) in orecs]
    assert names == onames, (tag, b, "dict order", names, onames)
    for (nm, sp, row, rw, te, tr), (_, _, orw, ote, otr) in zip(recs, orecs):
        assert np.float64(rw).tobytes() == np.float64(orw).tobytes(), (tag, b, nm, "reward", rw, orw)
        assert te == bool(ote) and tr == bool(otr), (tag, b, nm, "flags")
    es = tables["env_state"][b]
    fl = int(es[_abi.ENV_FLAGS])
    assert (bool(fl & _abi.ENVF_TERM_ALL), bool(fl & _abi.ENVF_TRUNC_ALL)) == (ota, otra), (tag, b, "__all__")
    assert int(es[_abi.ENV_STEP]) == orc.current_step, (tag, b, "current_step")
    got_next = (int(es[_abi.ENV_NEXT_PRED_ID]), int(es[_abi.ENV_NEXT_PRED_ID_T2]), int(es[_abi.ENV_NEXT_PREY_ID]),
                int(es[_abi.ENV_NEXT_PREY_ID_T2]))
    assert got_next == orc.next_ids, (tag, b, "next ids", got_next, orc.next_ids)
    assert int(es[_abi.ENV_N_PRED_ALIVE]) == orc.active_num_predators, (tag, b)
    assert int(es[_abi.ENV_N_PREY_ALIVE]) == orc.active_num_prey, (tag, b)
    if not (fl & (_abi.ENVF_WAS_RESET | _abi.ENVF_TRUNC_ALL)):
        assert int(es[_abi.ENV_DRAWS]) == int(orc._out.draws), (tag, b, "draws", int(es[_abi.ENV_DRAWS]), int(orc._out.draws))
    for (nm, sp, row, _, te, _) in recs:
        if te:
            continue
        s = env.pred_capacity * sp + row
        st = orc.agent_state(nm)
        xy = int(tables["row_xy"][b][s])
        assert (xy >> 8, xy & 255) == st["pos"], (tag, b, nm, "pos")
        assert np.float64(tables["row_energy"][b][s]).tobytes() == np.float64(st["energy"]).tobytes(), (tag, b, nm, "energy")
        assert np.float64(tables["row_cumrew"][b][s]).tobytes() == np.float64(st["cumulative_reward"]).tobytes(), (tag, b, nm, "cum")
        assert int(tables["row_lastrep"][b][s]) == st["last_reproduction"], (tag, b, nm, "last_reproduction")
    gxy, ge = orc.grass_state()
    assert tables["grass_energy"][b][: env.n_grass].tobytes() == ge.tobytes(), (tag, b, "grass energy")
    if env.walls:
        assert infos_of(env, b, recs, tables) == orc.infos_of_last_call(), (tag, b, "infos")
    op = env.obs_pred[b].cpu().numpy()
    oq = env.obs_prey[b].cpu().numpy()
    for k, (nm, sp, row, _, _, _) in enumerate(recs):
        want = orc.last_obs(k)
        got = (oq if sp else op)[row]
        assert got.astype(np.float32).tobytes() == want.tobytes(), (tag, b, nm, "obs")


def rollout_vs_oracle(env: BatchedRedQueen, make_oracle, seed0, n_calls, check_every=1, envs=None, check_grid=False):
    """Step `env` n_calls times with device-side random actions, device-side reproduction uniforms and auto-reset
    and check it against one oracle per env (same Philox contract).  The first call is the reset."""
    B = env.batch_size
    envs = list(range(B)) if envs is None else envs
    oracles = {b: make_oracle() for b in envs}
    env.set_seeds(seed0)
    env.env_state.zero_()
    env.env_state[:, _abi.ENV_FLAGS] = _abi.ENVF_DONE
    env.env_state[:, _abi.ENV_EPISODE] = -1
    n_resets = 0
    stats = dict(births=0, type2=0)
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
                    assert grid[b].astype(np.float32).tobytes() == oracles[b].grid_world_state.tobytes(), (t, b, "grid")
                es = tables["env_state"][b]
                n_resets += bool(int(es[_abi.ENV_FLAGS]) & _abi.ENVF_WAS_RESET)
                stats["births"] += int(es[_abi.ENV_N_PRED_NEW]) + int(es[_abi.ENV_N_PREY_NEW])
                nP, nQ = int(es[_abi.ENV_N_PRED_ROWS]), int(es[_abi.ENV_N_PREY_ROWS])
                _, t2, _ = split_row_id(tables["row_id"][b])
                stats["type2"] += int(t2[:nP].sum() + t2[env.pred_capacity: env.pred_capacity + nQ].sum())
    return n_resets, stats
