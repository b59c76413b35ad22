#!/usr/bin/env python3
"""Training-step time of asr_en_base on a fixed batch (no augmentation) for one build of the library, with the
per-family kernel times from the library's dispatch timestamps.  A/B tool: tools/step_time.py --lib build/variants/lib_X.so"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--T", type=int, default=1024)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
import torch
import bench
from voice100_amd import functional as F_, _native as N
if args.lib:
    N.LIB_PATH = os.path.abspath(args.lib)
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep

dev = torch.device("cuda:0")
N.load(); F_.set_matmul_precision("bf16")
torch.manual_seed(1234)
model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
model.batch_augment.forward = lambda a, l: (a, l)
step = TrainStep(model)
(audio, audio_len), tgt = bench.synth_batch(dev, 32, 1234)
audio = (audio[:, :args.T] if args.T <= audio.shape[1] else torch.cat([audio, audio[:, :args.T - audio.shape[1]]], 1)).contiguous()
batch = ((audio, torch.full_like(audio_len, args.T)), tgt)
for _ in range(5): step(batch)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(args.iters): step(batch)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / args.iters)
N.timing_enable(True)
for _ in range(5): step(batch)
torch.cuda.synchronize()
kt = {k: round(v[1] / 5, 3) for k, v in sorted(N.timing_read().items())}
N.timing_enable(False)
print(f"{os.path.basename(args.lib or 'default'):24s} T={args.T}: {best * 1e3:.3f} ms/step  {kt}")
