#!/usr/bin/env python3
"""Loss trajectory of the bench workload (asr_en_base, B=32 x 1024, seeded like bench.py) for a few precision / storage
settings: fp32, bf16 operands with fp32 storage, bf16 operands with bf16 hidden storage (act16 levels 2, 3 and 4 = the bench's).  The curves must track
each other (same data, same augmentation and dropout draws); used to check that reduced-precision storage trains the same.
python tools/loss_curve.py [--steps 40]"""
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


def run(precision, act16, steps, dev):
    F_.set_matmul_precision(precision)
    F_.set_activation_storage(act16)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
    step = TrainStep(model)
    batch = bench.synth_batch(dev, 32, 1234)
    return [float(step(batch)) for _ in range(steps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
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
