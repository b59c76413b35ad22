#!/usr/bin/env python3
"""Device WORLD analysis against the float64 oracle over many random speech-like signals of random lengths (ragged batches):
how often a discrete decision (voiced / unvoiced, love-train pass) differs, and the value errors where it does not.
python tools/fuzz_world_analysis.py [--n 48] [--fs 16000]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools.bench_world_analysis import speechlike


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    from voice100_amd.vocoder import WORLDVocoder
    from oracle import world_analysis as wa
    fs = args.fs
    v = WORLDVocoder(sample_rate=fs).cuda()
    rng = np.random.default_rng(123)
    tot = dict(frames=0, vuv=0, lt=0, f0=0.0, sp=0.0, ap=0.0, cod=0.0)
    for b0 in range(0, args.n, args.batch):
        xs = []
        for i in range(b0, min(args.n, b0 + args.batch)):
            x = speechlike(float(rng.uniform(0.4, 3.0)), fs, 1000 + i)
            if i % 5 == 0:
                x[: len(x) // 3] = 0                         # a stretch of digital silence
            if i % 7 == 0:
                x = (x * 30).clip(-1, 1).astype(np.float32)   # clipping
            xs.append(x)
        L = max(len(x) for x in xs)
        batch = torch.zeros((len(xs), L))
        for i, x in enumerate(xs):
            batch[i, :len(x)] = torch.from_numpy(x)
        lengths = torch.tensor([len(x) for x in xs], dtype=torch.int32)
        g0 = v.dio(batch.cuda(), lengths, f0_floor=80.0, f0_ceil=400.0)
        gs = v.cheaptrick(batch.cuda(), g0, lengths).cpu().numpy()
        ga, gc = v.d4c(batch.cuda(), g0, lengths)
        g0, ga, gc = g0.cpu().numpy(), ga.cpu().numpy(), gc.cpu().numpy()
        for i, x in enumerate(xs):
            xd = x.astype(np.float64)
            f0, tp = wa.dio(xd, fs, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
            T = len(f0)
            mism = (g0[i, :T] > 0) != (f0 > 0)
            tot["frames"] += T
            d = np.abs(g0[i, :T] - f0)
            if mism.any() or d[~mism].max() > 1e-6:
                idx = b0 + i
                where = np.nonzero(mism | (d > 1e-6))[0]
                print(f"  utterance {idx} ({'zeroed first third, ' if idx % 5 == 0 else ''}{'clipped, ' if idx % 7 == 0 else ''}{T} frames): frames {where.tolist()} "
                      f"device {np.round(g0[i, where], 2).tolist()} oracle {np.round(f0[where], 2).tolist()}; silence ends at frame {len(x) // 3 // (fs // 100) if idx % 5 == 0 else None}")
            tot["vuv"] += int(mism.sum())
            tot["f0"] = max(tot["f0"], float(np.abs(g0[i, :T] - f0)[~mism].max()))
            # the later stages on the DEVICE's f0 (so a voicing difference does not propagate into these comparisons)
            sp = wa.cheaptrick(xd, g0[i, :T], tp, fs, fft_size=v.n_fft)
            apo = wa.d4c(xd, g0[i, :T], tp, fs, fft_size=v.n_fft)
            tot["sp"] = max(tot["sp"], float(np.abs(np.log(gs[i, :T]) - np.log(sp)).max()))
            unv_o, unv_g = np.isclose(apo[:, 0], 1 - 1e-12), np.isclose(ga[i, :T, 0], 1 - 1e-12)
            lt = unv_o != unv_g
            tot["lt"] += int(lt.sum())
            tot["ap"] = max(tot["ap"], float(np.abs(ga[i, :T] - apo)[~lt].max()))
            tot["cod"] = max(tot["cod"], float(np.abs(gc[i, :T] - wa.code_aperiodicity(apo, fs))[~lt].max()))
    print(f"{args.n} utterances at {fs} Hz, {tot['frames']} frames: voicing mismatches {tot['vuv']}, love-train mismatches {tot['lt']}; "
          f"max |f0 diff| {tot['f0']:.3e} Hz, max |log sp diff| {tot['sp']:.3e}, max |ap diff| {tot['ap']:.3e}, max |coded diff| {tot['cod']:.3e} dB")


if __name__ == "__main__":
    main()
