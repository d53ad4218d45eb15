"""GPU experiment: do the placements a process draws get better after it has allocated and freed device memory once?
Builds three spread candidates and probes them, frees everything, builds three more, ... (first thing on a fresh box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from predpreygrass_amd.config import config_env  # noqa: E402
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass  # noqa: E402

spread = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print("gpu", torch.cuda.get_device_properties(0).uuid, torch.cuda.get_device_name(0))
for rnd in range(4):
    t0 = time.perf_counter()
    g = SubBatchedPredPreyGrass(dict(config_env), batch_size=4096, n_sub=3, device="cuda:0", obs_dtype=torch.float64, obs_spread=spread,
                                placement_candidates=3)
    print(f"round {rnd}: candidates {[round(v, 1) for v in g.placement_probe_us]} us per step  ({time.perf_counter() - t0:.1f} s)", flush=True)
    for e in g.subs:
        e.close()
    del g
    torch.cuda.synchronize()
    if rnd == 1:
        torch.cuda.empty_cache()
        print("(torch.cuda.empty_cache())")
