"""Diagnostic (experiment builds, -DPPG_EXPERIMENTS): per-tile phase stamps of one policy step written by ppg_policy_forward_*
(env PPG_POLICY_TIMELINE=<prefix>) -> where the launch's time goes.   usage: python tools/policy_timeline.py <prefix>"""
import sys

import numpy as np

prefix = sys.argv[1]
rows = {}
for sp in ("prey", "pred"):
    try:
        a = np.fromfile(f"{prefix}.{sp}", dtype=np.uint64)
    except FileNotFoundError:
        continue
    a = a.reshape(-1, 8 if "--words8" in sys.argv else 64)
    a = a[a[:, 3] > 0]
    rows[sp] = a
t0 = min(int(a[:, 3].min()) for a in rows.values())
for sp, a in rows.items():
    wg, hw, ns = a[:, 0].astype(int), a[:, 1], a[:, 2].astype(int)
    st = (a[:, 3:7].astype(np.int64) - t0) / 100.0      # us
    xcc = (hw >> np.uint64(32)).astype(int) & 0xF
    cu = ((hw.astype(np.int64) >> 8) & 0xF) | (((hw.astype(np.int64) >> 13) & 0x7) << 4) | (xcc << 8)
    print(f"== {sp}: {len(a)} tiles on {len(set(wg))} workgroups, {len(set(cu))} distinct (xcc, se, cu); samples {ns.sum()}")
    print(f"   first tile starts {st[:, 0].min():8.1f} us, last tile ends {st[:, 3].max():8.1f} us")
    full = ns == 128
    for name, m in (("full tiles", full), ("small tiles", ~full)):
        if not m.any():
            continue
        d = st[m]
        conv, fc1, head = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
        print(f"   {name:11s} n={m.sum():5d} samples/tile {ns[m].mean():6.1f}   conv {conv.mean():7.1f} us (p10 {np.percentile(conv,10):6.1f} p90 {np.percentile(conv,90):6.1f})"
              f"   fc1 {fc1.mean():7.1f} (p10 {np.percentile(fc1,10):6.1f} p90 {np.percentile(fc1,90):6.1f})   head {head.mean():6.1f}   tile {(d[:,3]-d[:,0]).mean():7.1f}")
    # per workgroup: busy time and gaps
    ends = {}
    for w in set(wg):
        m = wg == w
        ends[w] = (st[m, 0].min(), st[m, 3].max(), (st[m, 3] - st[m, 0]).sum())
    e = np.array(list(ends.values()))
    print(f"   workgroups: start p50 {np.percentile(e[:,0],50):7.1f} p90 {np.percentile(e[:,0],90):7.1f} max {e[:,0].max():7.1f};  end p10 {np.percentile(e[:,1],10):7.1f} p50 {np.percentile(e[:,1],50):7.1f} max {e[:,1].max():7.1f};  busy/span {np.mean(e[:,2]/(e[:,1]-e[:,0])):.3f}")
    # occupancy over time
    T = st[:, 3].max()
    grid = np.linspace(0, T, 11)
    occ = [int(((st[:, 0] <= t) & (st[:, 3] > t)).sum()) for t in grid]
    print("   tiles in flight at 0,10,..100 % of the span:", occ)
    rounds = np.zeros(len(a), dtype=int)
    for w in set(wg):
        idx = np.where(wg == w)[0]
        rounds[idx[np.argsort(st[idx, 0])]] = np.arange(len(idx))
    for r in range(rounds.max() + 1):
        m = rounds == r
        print(f"   round {r}: {m.sum():4d} tiles, start {st[m,0].mean():7.1f}, tile time {(st[m,3]-st[m,0]).mean():7.1f} us, samples {ns[m].mean():6.1f}")

    if a.shape[1] == 64:   # per-wave cycle counters of the phases (s_memtime)
        m = ns == ns.max()
        w = a[m][:, 8:56].astype(np.int64).reshape(-1, 4, 12)
        names = ["start-up + drain", "staging (+ load wait)", "(unused)", "conv1", "barrier", "conv2", "barrier", "obs request", "conv3", "barrier",
                 "fc1 vmcnt wait", "fc1 barrier"]
        rest = a[m][:, 56:60].astype(np.int64)
        tot = w.sum(axis=2) + rest
        print(f"   per-wave cycles of a tile of {ns.max()} samples (mean over {m.sum()} tiles and 4 waves; wave 0..3 means in brackets); total {tot.mean():.0f}")
        for i, nme in enumerate(names):
            print(f"     {nme:20s} {w[:, :, i].mean():9.0f}  {100 * w[:, :, i].mean() / tot.mean():5.1f} %   [{', '.join(f'{w[:, k, i].mean():.0f}' for k in range(4))}]")
        print(f"     {'fc1 issue + compute':20s} {rest.mean():9.0f}  {100 * rest.mean() / tot.mean():5.1f} %   [{', '.join(f'{rest[:, k].mean():.0f}' for k in range(4))}]")
