"""predpreygrass_amd -- a batched Predator-Prey-Grass environment whose transition runs as
hand-written HIP on AMD MI355X (gfx950).

    from predpreygrass_amd import PredPreyGrass, config_env        # reference-shaped dict API
    from predpreygrass_amd import BatchedPredPreyGrass             # tensor API, B envs per GPU
    from predpreygrass_amd.red_queen import PredPreyGrass, BatchedRedQueen, config_env_base   # second generation
"""
from .config import config_env, resolve_config  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch / a GPU
    if name == "BatchedPredPreyGrass":
        from .batched import BatchedPredPreyGrass
        return BatchedPredPreyGrass
    if name in ("PredPreyGrass", "env_creator", "VectorPredPreyGrass"):
        from . import env
        return getattr(env, name)
    if name == "BatchedRedQueen":
        from .red_queen import BatchedRedQueen
        return BatchedRedQueen
    if name in ("PredPreyGrassParallelEnv", "PredPreyGrassAECEnv", "parallel_env"):
        from . import pettingzoo_env
        return getattr(pettingzoo_env, name)
    raise AttributeError(name)
