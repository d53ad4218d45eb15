#!/usr/bin/env python3
"""Calibration ("confirm with a device copy benchmark on the box"): achievable HBM
write and copy rates on this GPU for buffers the size of one step's observation traffic and larger."""
import sys, torch
dev = "cuda:0"
def rate(fn, nbytes, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
for mb in (128, 384, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev); b = torch.empty(n, dtype=torch.float64, device=dev)
    w = rate(lambda: a.fill_(0.0), n * 8)
    c = rate(lambda: b.copy_(a), 2 * n * 8)
    print(f"{mb:5d} MiB: fill (write-only) {w:.2f} TB/s   copy (read+write) {c:.2f} TB/s", flush=True)
