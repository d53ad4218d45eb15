"""Import shim for the *Python reference* environment (test infrastructure only).

THIS FILE IS TEST INFRASTRUCTURE.  It is used in the build container only, to
(1) validate the C restatement in ``oracle/ppg_oracle.c`` and (2) generate the
golden vectors under ``tests/golden/``.  Nothing in the product package
(``predpreygrass_amd/``) imports it, and it is never needed on the GPU box
(``/root/reference`` does not exist there).

It copies no reference code: it reads
``/root/reference/predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py``
at run time, applies a purely syntactic rewrite that makes the file parse on
Python 3.10 (the reference needs >= 3.11 for star-unpacking inside a subscript,
e.g. ``grid[1, *pos]`` at predpreygrass_rllib_env.py:195), stubs the two
third-party imports that are absent in this image (``gymnasium`` :12 and
``ray.rllib`` :13-14) and ``exec``s the result.
"""
from __future__ import annotations

import os
import re
import sys
import types
import typing

import numpy as np

REFERENCE_ROOT = os.environ.get("PPG_REFERENCE_ROOT", "/root/reference")
BASE_ENV_RELPATH = "predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py"
# base-family variants whose step() is the base step() plus a small delta
VARIANTS = {
    "base": "predpreygrass/non_evolutionary/base_environment",
    "seasonal": "predpreygrass/non_evolutionary/base_environment_seasonal",
    "sparse_rewards": "predpreygrass/non_evolutionary/project_reward_shaping/base_environment_sparse_rewards",
    "sparse_rewards_plus_eating": "predpreygrass/non_evolutionary/project_reward_shaping/base_environment_sparse_rewards_plus_eating",
    "dense_rewards": "predpreygrass/non_evolutionary/project_reward_shaping/base_environment_dense_rewards",
    "dense_rewards_additive": "predpreygrass/non_evolutionary/project_reward_shaping/base_environment_dense_rewards_additive",
    "sparse_rewards_plus_kickback": "predpreygrass/non_evolutionary/project_reward_shaping/base_environment_sparse_rewards_plus_kickback",
    "drive_conditioned": "predpreygrass/non_evolutionary/drive_conditioned_environment",
    # second generation without walls / line of sight; its configs live in config/*.py
    "red_queen": "predpreygrass/non_evolutionary/red_queen",
    # the same plus static walls, line-of-sight masking and per-agent infos
    "walls_occlusion": "predpreygrass/non_evolutionary/walls_occlusion",
}


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, BASE_ENV_RELPATH))


def _install_stub_modules() -> None:
    """Minimal stand-ins for `gymnasium.spaces` and `ray.rllib` (import-time only)."""
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")

        class Box:  # attribute bag; the env only constructs and stores it
            def __init__(self, low, high, shape=None, dtype=np.float64):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        class Discrete:
            def __init__(self, n):
                self.n = int(n)

            def sample(self):
                return int(np.random.randint(self.n))

        spaces.Box, spaces.Discrete = Box, Discrete
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces

    if "ray" not in sys.modules:
        names = [
            "ray",
            "ray.rllib",
            "ray.rllib.env",
            "ray.rllib.env.multi_agent_env",
            "ray.rllib.utils",
            "ray.rllib.utils.typing",
        ]
        mods = {n: types.ModuleType(n) for n in names}

        class MultiAgentEnv:
            def __init__(self, *a, **k):
                pass

            def reset(self, *, seed=None, options=None):
                return None

            def close(self):
                pass

        mods["ray.rllib.env.multi_agent_env"].MultiAgentEnv = MultiAgentEnv
        t = mods["ray.rllib.utils.typing"]
        t.AgentID = typing.Any
        t.Dict, t.List, t.Tuple = typing.Dict, typing.List, typing.Tuple
        for n, m in mods.items():
            sys.modules[n] = m
            if "." in n:
                parent, child = n.rsplit(".", 1)
                setattr(mods[parent], child, m)


_STAR_SUBSCRIPT = re.compile(r"\[(\w+), \*")


def _rewrite_star_subscripts(text: str) -> tuple[str, int]:
    """``a[k, *expr]`` -> ``a[(k, *expr)]`` with bracket matching (py3.10 syntax)."""
    out = []
    pos = 0
    count = 0
    while True:
        m = _STAR_SUBSCRIPT.search(text, pos)
        if m is None:
            out.append(text[pos:])
            break
        # walk forward from the opening '[' to its matching ']'
        depth = 0
        i = m.start()
        while True:
            ch = text[i]
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        out.append(text[pos : m.start()])
        out.append("[(" + text[m.start() + 1 : i] + ")]")
        pos = i + 1
        count += 1
    return "".join(out), count


_cached_modules = {}


def load_reference_module(variant: str = "base"):
    """Return a module object holding the reference `PredPreyGrass` class of a base-family variant."""
    if variant in _cached_modules:
        return _cached_modules[variant]
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_stub_modules()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    path = os.path.join(REFERENCE_ROOT, VARIANTS[variant], "predpreygrass_rllib_env.py")
    with open(path, "r") as fh:
        text = fh.read()
    text, n = _rewrite_star_subscripts(text)
    if n == 0:
        raise RuntimeError("star-subscript rewrite found nothing; reference layout changed?")
    mod = types.ModuleType("_ppg_reference_env_" + variant)
    mod.__file__ = path
    exec(compile(text, path, "exec"), mod.__dict__)
    _cached_modules[variant] = mod
    return mod


def reference_default_config(variant: str = "base") -> dict:
    if variant in ("red_queen", "walls_occlusion"):  # these envs take an explicit config; config/config_env_base.py is their base
        return reference_config(f"predpreygrass.non_evolutionary.{variant}.config.config_env_base", "config_env_base")
    mod = load_reference_module(variant)
    return dict(mod.config_env)  # each variant file imports its own config_env (line 5)


def reference_config(module: str, attr: str = "config_env") -> dict:
    """A config dict defined by one of the reference's config modules (plain data, imported as is)."""
    import importlib

    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return dict(getattr(importlib.import_module(module), attr))


def make_reference_env(overrides: dict | None = None, variant: str = "base"):
    """`PredPreyGrass(config)` of the reference with the variant's `config_env` defaults + overrides."""
    mod = load_reference_module(variant)
    cfg = reference_default_config(variant)
    if overrides:
        cfg.update(overrides)
    return mod.PredPreyGrass(cfg)
