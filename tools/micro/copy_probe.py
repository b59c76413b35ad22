#!/usr/bin/env python3
"""Tune / report the device-copy yardstick: torch's copy_ against the library's v100_copy_probe variants (1 GiB -> 1 GiB).
Each variant runs in a child process (the variant is read once from V100_COPY_VARIANT).  python tools/micro/copy_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from voice100_amd import _native as N
dev = torch.device("cuda:0")
src = torch.empty(1 << 28, device=dev, dtype=torch.float32).normal_()
dst = torch.empty_like(src)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
print("%%s torch.copy_ %%.0f GB/s   v100_copy_probe %%.0f GB/s" %% (sys.argv[1], t(lambda: dst.copy_(src)), t(lambda: N.call("v100_copy_probe", src, dst, src.numel() * 4))))
assert torch.equal(src, dst)
''' % ROOT
for variant in ("101", "102", "103", "111", "112", "401", "402", "801", "802", "811", "812", "1601", "1602", "3201", "3202"):
    env = dict(os.environ, V100_COPY_VARIANT=variant)
    subprocess.run([sys.executable, "-c", CHILD, variant], env=env, check=False)
