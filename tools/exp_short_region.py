"""GPU experiment: the driver's timed region (sync, 20 steps on 3 sub-batch streams, sync) repeated many times in ONE process:
distribution of the wall time and of the event time, for a few stream counts.  usage: python tools/exp_short_region.py [reps=40]"""
import sys, time, statistics
import torch
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
for n_sub in (3, 2, 4, 1):
    g = SubBatchedPredPreyGrass({}, batch_size=4096, n_sub=n_sub, device=dev, obs_dtype=torch.float64, seed=0)
    g.reset()
    for i in range(3072):
        if i % 64 == 0:
            g.rebalance()
        g.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    walls, evs = [], []
    for r in range(reps):
        for _ in range(5):
            g.step(random_actions=True, auto_reset=True)
        torch.cuda.synchronize()
        time.sleep(0.002 * (r % 3))     # (the driver's run has host work between the warm-up and the timed steps)
        e0 = [torch.cuda.Event(enable_timing=True) for _ in g.streams]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in g.streams]
        t0 = time.perf_counter()
        for s, e in zip(g.streams, e0):
            e.record(s)
        for _ in range(20):
            g.step(random_actions=True, auto_reset=True)
        for s, e in zip(g.streams, e1):
            e.record(s)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e6 / 20)
        evs.append(sum(a.elapsed_time(b) for a, b in zip(e0, e1)) / len(e0) * 1e3 / 20)
    print(f"n_sub {n_sub}: wall/step us median {statistics.median(walls):.1f} min {min(walls):.1f} max {max(walls):.1f} | events/step median "
          f"{statistics.median(evs):.1f} min {min(evs):.1f} max {max(evs):.1f}")
    print("   walls:", " ".join(f"{w:.0f}" for w in walls))
    del g
