#!/usr/bin/env python3
"""Loss trajectory of the bench workload (asr_en_base, B=32 x 1024, seeded like bench.py) for a few precision / storage
settings: fp32, bf16 operands with fp32 storage, bf16 operands with bf16 hidden storage (act16 levels 2, 3 and 4 = the bench's).  The curves must track
each other (same data, same augmentation and dropout draws); used to check that reduced-precision storage trains the same.
python tools/loss_curve.py [--steps 40]
python tools/loss_curve.py --task learnable --steps 600     (review round 3, item 6: >= 500 steps on a task that CAN be learnt --
    the targets are a deterministic function of the input: every token is rendered as a 10-frame run of its own fixed 64-bin
    pattern plus noise, a fresh batch per step from a per-step seed -- fp32 against bf16 / activation storage level 4, same
    batches, same augmentation and dropout draws; the loss must FALL, and fall the same way)"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from voice100_amd import functional as F_
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep


def learnable_batch(dev, step_idx, table, B=32, L=100, run=10, T=1024):
    """Utterance = L tokens, each rendered as `run` frames of its pattern table[token] (+ N(0, 0.5) noise), tail = the blank pattern:
    the transcript is a deterministic function of the audio, so CTC can be driven towards 0."""
    g = torch.Generator().manual_seed(100000 + step_idx)
    text = torch.randint(1, 29, (B, L), generator=g)
    frames = table[text].repeat_interleave(run, dim=1)                         # [B, L * run, 64]
    audio = table[0].expand(B, T, 64).clone()
    audio[:, :L * run] = frames
    audio = audio + 0.5 * torch.randn(B, T, 64, generator=g)
    return ((audio.to(dev), torch.full((B,), T, dtype=torch.int32, device=dev)),
            (text.to(dev), torch.full((B,), L, dtype=torch.int32, device=dev)))


def run(precision, act16, steps, dev, task="bench"):
    F_.set_matmul_precision(precision)
    F_.set_activation_storage(act16)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
    step = TrainStep(model)
    if task == "bench":
        batch = bench.synth_batch(dev, 32, 1234)
        return [float(step(batch)) for _ in range(steps)]
    table = torch.randn(29, 64, generator=torch.Generator().manual_seed(4321)) * 2 - 4
    return [float(step(learnable_batch(dev, i, table))) for i in range(steps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--task", default="bench", choices=["bench", "learnable"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.task == "learnable":
        curves = {"fp32": run("fp32", 0, args.steps, dev, "learnable"), "bf16/act4": run("bf16", 4, args.steps, dev, "learnable"),
                  "bf16/act5": run("bf16", 5, args.steps, dev, "learnable")}
        print(f"# learnable synthetic task (tools/loss_curve.py --task learnable), {args.steps} steps, asr_en_base, B = 32 x 1024 frames, "
              "augmentation + dropout on, Adam 1e-3; mean CTC loss per window of 20 steps")
        print("steps      " + " ".join(f"{k:>10s}" for k in curves))
        w = 20
        for i in range(0, args.steps, w):
            print(f"{i:4d}-{min(i + w, args.steps) - 1:4d}  " + " ".join(f"{np.mean(v[i:i + w]):10.4f}" for v in curves.values()))
        tail = slice(max(0, args.steps - 100), args.steps)
        for name, v in curves.items():
            a = np.array(v)
            print(f"{name:10s}: first-20 mean {a[:20].mean():.3f}, last-100 mean {a[tail].mean():.4f}, first step with loss < 1.0: "
                  f"{int(np.argmax(a < 1.0)) if (a < 1.0).any() else None}")
        return
    curves = {"fp32": run("fp32", 0, args.steps, dev), "bf16/act0": run("bf16", 0, args.steps, dev),
              "bf16/act2": run("bf16", 2, args.steps, dev), "bf16/act3": run("bf16", 3, args.steps, dev),
              "bf16/act4": run("bf16", 4, args.steps, dev)}          # level 4 = what bench.py runs
    print("step " + " ".join(f"{k:>10s}" for k in curves))
    for i in range(args.steps):
        print(f"{i:4d} " + " ".join(f"{v[i]:10.4f}" for v in curves.values()))
    ref = np.array(curves["fp32"])
    for k, v in curves.items():
        d = np.abs(np.array(v) - ref) / ref
        print(f"{k}: max rel deviation from fp32 over steps 0-9 {d[:10].max():.4f}, over all {d.max():.4f}, final loss {v[-1]:.4f}")


if __name__ == "__main__":
    main()
