"""GPU experiment: mean agent rows per env in 64-step windows over a long run of the headline workload (how long does the
population take to become stationary after the synchronous reset?)."""
import sys
import torch
sys.path.insert(0, ".")
from predpreygrass_amd import _abi
from predpreygrass_amd.config import config_env
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = SubBatchedPredPreyGrass(dict(config_env), batch_size=4096, n_sub=3, device="cuda:0", seed=0)
g.reset()
out = []
for w in range(n // 64):
    torch.cuda.synchronize()
    for e in g.subs:
        e.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1] = 0
    for _ in range(64):
        g.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    rows = sum(int(e.env_state[:, _abi.ENV_OBS_PRED:_abi.ENV_OBS_PREY + 1].sum().item()) for e in g.subs)
    out.append(round(rows / 4096 / 64, 2))
print(out)
import numpy as np
a = np.array(out)
print("mean of windows 32..:", a[32:].mean(), "min/max", a[32:].min(), a[32:].max())
