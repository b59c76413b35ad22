#!/usr/bin/env python3
"""WORLD synthesis on the device at the configs[2] size: B = 16 utterances x 1023 frames (10.2 s each at 16 kHz, 10 ms frames),
mixed voicing.  python tools/bench_world.py [--iters 20]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--T", type=int, default=1023)
    args = ap.parse_args()
    from voice100_amd.vocoder import WORLDVocoder
    dev = torch.device("cuda")
    v = WORLDVocoder()
    rng = np.random.RandomState(5)
    B, T = args.B, args.T
    f0 = np.where(np.sin(np.arange(T)[None] / 40.0 + rng.rand(B, 1) * 6) > 0.2, 0.0,
                  90 + 160 * rng.rand(B, 1) + 20 * np.sin(np.arange(T)[None] / 7.0)).astype(np.float32)
    k = np.arange(257)
    sp = (1e-2 * (1 + 4 * np.exp(-((k * 16000 / 512 - 1500) / 300.0) ** 2))[None, None] * np.exp(0.3 * rng.randn(B, T, 1))).astype(np.float32)
    cod = np.where(f0[..., None] > 0, -10 - 25 * rng.rand(B, T, 1), 0.0).astype(np.float32)
    f0, sp, cod = (torch.from_numpy(a).to(dev) for a in (f0, sp, cod))
    ap_ = v.decode_aperiodicity(cod)
    for _ in range(3):
        y, n = v.synthesize(f0, sp, ap_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        ap_ = v.decode_aperiodicity(cod)
        y, n = v.synthesize(f0, sp, ap_)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    secs = B * y.shape[1] / 16000.0
    print(f"WORLD synthesis B={B} x {T} frames: {dt*1e3:.3f} ms per batch, {int(n.sum())} pulses, {secs:.1f} s of audio -> {secs/dt:.0f} x real time, "
          f"{B*T/dt/1e6:.2f} M WORLD frames/s")


if __name__ == "__main__":
    main()
