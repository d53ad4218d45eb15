#!/usr/bin/env python3
"""Generate the ANALYTICS golden vectors (tests/golden/rq/*.analytics.json.gz, tests/golden/wo/*.analytics.json.gz) FROM THE
REFERENCE ITSELF: the per-agent bookkeeping the second-generation envs keep beside their transition (RQ:99-116:
unique_agents, unique_agent_stats, death_agents_stats, per_step_agent_data, agent_ages, agent_parents, offspring lists,
agent_activation_counts).

Runs only in the build container (needs /root/reference).  The episode is the one of the existing golden case of the same name:
the reference env is reset with the case's seed and driven with the case's recorded action dicts (tests/golden/rq/<name>.npz),
so the two files describe the same calls.  At the checkpoints (every `every` calls and the last call) the books are dumped as
JSON (Python floats round-trip exactly through repr); per_step_agent_data is kept for the checkpoint calls themselves.

    python tests/golden/make_golden_rq_analytics.py [case ...]
"""
from __future__ import annotations

import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.ref_shim import make_reference_env  # noqa: E402
from tests.golden_io_rq import RQGoldenCase  # noqa: E402

CASES = {   # name: checkpoint spacing
    "rq_mixed_types_seed7": 10,
    "rq_pool_exhaust_seed2": 10,
    "rq_shuffled_seed5": 9,
    "rq_base_seed3": 60,
    "wo_los_two_types_seed5": 10,
    "wo_mask_only_shuffled_seed6": 10,
}


def plain(o):
    """numpy scalars / tuples / nested containers -> what json stores (floats stay float64 values)."""
    if isinstance(o, dict):
        return {str(k): plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    if isinstance(o, np.ndarray):
        return [plain(v) for v in o.tolist()]
    if isinstance(o, (np.bool_, bool)):
        return bool(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    return o


def books(env, t):
    return {
        "call": t,
        "unique_agents": plain(env.unique_agents),
        "unique_agent_stats": plain(env.unique_agent_stats),
        "death_agents_stats": plain(env.death_agents_stats),
        "agent_ages": plain(env.agent_ages),
        "agent_parents": plain(env.agent_parents),
        "agent_offspring_counts": plain(env.agent_offspring_counts),
        "agent_live_offspring_ids": plain(env.agent_live_offspring_ids),
        "agent_activation_counts": plain({k: v for k, v in env.agent_activation_counts.items() if v}),
        "per_step_agent_data_len": len(env.per_step_agent_data),
        "per_step_agent_data_last": plain(env.per_step_agent_data[-1]) if env.per_step_agent_data else None,
    }


def make_case(name, every):
    case = RQGoldenCase(name)
    variant = "walls_occlusion" if case.walls else "red_queen"
    env = make_reference_env(dict(case.config), variant)
    obs, _ = env.reset(seed=int(case.z["seed"]))
    assert list(obs) == case.reset_keys
    points = [books(env, -1)]
    for t in range(case.n_calls):
        o, r, te, tr, infos = env.step(case.actions(t))
        assert [(k, float(r[k]), bool(te[k]), bool(tr[k])) for k in o] == case.records(t), (name, t, "not the recorded episode")
        if t % every == every - 1 or t == case.n_calls - 1:
            points.append(books(env, t))
    out_dir = os.path.join(HERE, "wo" if case.walls else "rq")
    path = os.path.join(out_dir, name + ".analytics.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps({"name": name, "points": points}).encode())
    last = points[-1]
    print(f"{name}: {case.n_calls} calls, {len(points)} checkpoints, {len(last['unique_agent_stats'])} agents ever, "
          f"{len(last['death_agents_stats'])} death entries, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    only = sys.argv[1:]
    for n, ev in CASES.items():
        if not only or n in only:
            make_case(n, ev)
