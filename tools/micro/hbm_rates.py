#!/usr/bin/env python3
"""Device-memory rates of the box with stock torch ops: fill (write only), copy (read + write), sum (read only)."""
import torch
dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for mb in (64, 256, 1024):
    n = mb * (1 << 20) // 4
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    a.normal_()
    dt = t(lambda: b.fill_(1.0)); print(f"{mb:5d} MiB fill : {mb*1.048576e6/dt/1e12:5.2f} TB/s written")
    dt = t(lambda: b.copy_(a)); print(f"{mb:5d} MiB copy : {2*mb*1.048576e6/dt/1e12:5.2f} TB/s read+written")
    dt = t(lambda: a.sum()); print(f"{mb:5d} MiB sum  : {mb*1.048576e6/dt/1e12:5.2f} TB/s read")
