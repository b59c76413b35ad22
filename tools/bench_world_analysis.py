#!/usr/bin/env python3
"""WORLD analysis on the device (WORLDVocoder.encode_batch: DIO + CheapTrick + D4C + coding) at the configs[2] size: B = 16 utterances
x 10.2 s at 16 kHz; the oracle (float64 numpy restatement, one core) timed on ONE of them beside it, and the device results of that
utterance compared with the oracle's.
python tools/bench_world_analysis.py [--iters 10]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def speechlike(seconds, fs, seed):
    """harmonic source with a wandering F0, voiced / unvoiced alternation, a formant-ish tilt and a noise floor"""
    rng = np.random.default_rng(seed)
    n = int(seconds * fs)
    t = np.arange(n) / fs
    f0 = 110 + 60 * rng.random() + 25 * np.sin(2 * np.pi * (0.4 + 0.3 * rng.random()) * t) + 10 * np.sin(2 * np.pi * 2.3 * t)
    ph = 2 * np.pi * np.cumsum(f0) / fs
    x = sum(np.cos(k * ph + rng.random() * 6) / k for k in range(1, 24)) * 0.08
    gate = (np.sin(2 * np.pi * (0.7 + 0.2 * rng.random()) * t + rng.random() * 6) > -0.3).astype(np.float64)
    gate = np.convolve(gate, np.hanning(801) / np.hanning(801).sum(), mode="same")
    return (x * gate + rng.standard_normal(n) * (2e-3 + 2e-2 * (1 - gate))).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.23)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    from voice100_amd.vocoder import WORLDVocoder
    fs = 16000
    v = WORLDVocoder().cuda()
    xs = np.stack([speechlike(args.seconds, fs, s) for s in range(args.B)])
    x = torch.from_numpy(xs).cuda()
    for _ in range(2):
        f0, feat, cod = v.encode_batch(x)
    torch.cuda.synchronize()
    parts = {}
    for name, fn in (("dio", lambda: v.dio(x, f0_floor=80.0, f0_ceil=400.0)),
                     ("cheaptrick", lambda: v.cheaptrick(x, f0.double(), log=True)),
                     ("d4c", lambda: v.d4c(x, f0.double(), coded_only=True))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        parts[name] = (time.perf_counter() - t0) / args.iters
    t0 = time.perf_counter()
    for _ in range(args.iters):
        f0, feat, cod = v.encode_batch(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    secs = args.B * args.seconds
    T = f0.shape[1]
    print(f"WORLD analysis B={args.B} x {args.seconds:.2f} s ({T} frames each): {dt*1e3:.2f} ms per batch "
          f"(dio {parts['dio']*1e3:.2f}, cheaptrick {parts['cheaptrick']*1e3:.2f}, d4c {parts['d4c']*1e3:.2f}) -> {secs/dt:.0f} x real time, "
          f"{args.B*T/dt/1e6:.3f} M WORLD frames/s; voiced frames {int((f0 > 0).sum())} of {f0.numel()}")
    if args.no_oracle:
        return
    from oracle import world_analysis as wa
    xd = xs[0].astype(np.float64)
    t0 = time.perf_counter()
    of0, tp = wa.dio(xd, fs, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
    t1 = time.perf_counter()
    osp = wa.cheaptrick(xd, of0, tp, fs, fft_size=512)
    t2 = time.perf_counter()
    oap = wa.d4c(xd, of0, tp, fs, fft_size=512)
    oc = wa.code_aperiodicity(oap, fs)
    t3 = time.perf_counter()
    print(f"oracle (numpy float64, 1 core), ONE utterance: dio {1e3*(t1-t0):.0f} ms, cheaptrick {1e3*(t2-t1):.0f} ms, d4c {1e3*(t3-t2):.0f} ms "
          f"-> {args.seconds/(t3-t0):.1f} x real time")
    g0 = v.dio(x[:1], f0_floor=80.0, f0_ceil=400.0)
    gs = v.cheaptrick(x[:1], g0)[0].cpu().numpy()
    ga, gc = v.d4c(x[:1], g0)
    g0 = g0[0].cpu().numpy()
    print(f"device vs oracle on it: voicing mismatches {int(((g0 > 0) != (of0 > 0)).sum())} of {len(of0)} frames, max |f0 diff| {np.abs(g0 - of0).max():.3e} Hz, "
          f"max |log sp diff| {np.abs(np.log(gs) - np.log(osp)).max():.3e}, max |ap diff| {np.abs(ga[0].cpu().numpy() - oap).max():.3e}, "
          f"max |coded ap diff| {np.abs(gc[0].cpu().numpy() - oc).max():.3e} dB")


if __name__ == "__main__":
    main()
