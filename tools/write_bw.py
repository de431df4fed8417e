#!/usr/bin/env python3
"""Pure-write / pure-read / copy bandwidth of this box with torch's vectorised kernels (reference points for the
HBM-bound kernels: the int8 GEMMs are write-dominated)."""
import torch, time
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (39, 156, 292, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tw = t(lambda: x.fill_(1.0))
    tr = t(lambda: x.sum())
    tc = t(lambda: y.copy_(x))
    print("%5d MB: fill %6.1f us %5.2f TB/s | sum %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s (r+w)" %
          (mb, tw * 1e6, n * 4 / tw / 1e12, tr * 1e6, n * 4 / tr / 1e12, tc * 1e6, 2 * n * 4 / tc / 1e12))
