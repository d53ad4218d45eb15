"""Diagnostic: where a PredPreyGrass.step(action_dict) call spends its host time (cProfile over 2000 calls, one env).  usage on a GPU box: python3 tools/prof_dict_api.py"""
import cProfile, pstats, sys, time, random
sys.path.insert(0, '.')
import torch
from predpreygrass_amd.env import PredPreyGrass
from predpreygrass_amd.config import config_env
env = PredPreyGrass(config_env, device="cuda:0")
obs, _ = env.reset(seed=1)
rng = random.Random(0)
live = list(obs)
def run(n):
    global obs, live
    for _ in range(n):
        act = {a: rng.randrange(9) for a in live}
        obs, rew, term, trunc, info = env.step(act)
        live = [a for a in obs if not term.get(a, False)]
        if term.get("__all__") or trunc.get("__all__"):
            obs, _ = env.reset(seed=rng.randrange(1 << 30))
            live = list(obs)
run(300)
t = time.perf_counter(); run(2000); dt = time.perf_counter() - t
print("us per call", dt / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable(); run(2000); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
