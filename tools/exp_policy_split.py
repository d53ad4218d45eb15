"""GPU experiment helper (not product code): where a policy step's time goes -- prey network alone, predator network alone, both, and the
plan launches -- on one frozen env state (4096 envs after a random-action pre-roll, bf16 rows).   python tools/exp_policy_split.py [arch kwargs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from predpreygrass_amd import _abi  # noqa: E402
from predpreygrass_amd.batched import BatchedPredPreyGrass  # noqa: E402
from predpreygrass_amd.config import config_env  # noqa: E402
from predpreygrass_amd.policy import FusedPolicy, PolicyNet  # noqa: E402

env = BatchedPredPreyGrass(dict(config_env), batch_size=4096, device="cuda:0", obs_dtype=torch.bfloat16, seed=1)
env.reset()
for _ in range(600):
    env.step(random_actions=True, auto_reset=True)
torch.cuda.synchronize()
es = env.env_state.cpu().numpy()
print("rows in use: predators", int(es[:, _abi.ENV_N_PRED_ROWS].sum()), "prey", int(es[:, _abi.ENV_N_PREY_ROWS].sum()))
torch.manual_seed(0)
nets = [PolicyNet(env.Rp), PolicyNet(env.Rq)]
for name, pair in (("both", (nets[0], nets[1])), ("prey only", (None, nets[1])), ("predators only", (nets[0], None))):
    fused = FusedPolicy(*pair)
    for _ in range(10):
        fused.act(env, sample=True, seed=3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    t0 = time.perf_counter()
    e0.record()
    for i in range(n):
        fused.act(env, sample=True, seed=i)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:16s}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per act() on the device, {(time.perf_counter() - t0) / n * 1e6:8.1f} us wall")
    fused.close()
