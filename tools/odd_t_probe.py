#!/usr/bin/env python3
"""Step time of the asr_en_base training step at fixed input lengths (no augmentation): what row alignment costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voice100_amd import functional as F_, _native as N
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep

dev = torch.device("cuda:0")
N.load(); F_.set_matmul_precision("bf16")
torch.manual_seed(1234)
model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
model.batch_augment.forward = lambda a, l: (a, l)
step = TrainStep(model)
for T in (1024, 1023, 1022, 1020, 1016, 768, 767, 1279, 1280):
    (audio, audio_len), tgt = bench.synth_batch(dev, 32, 1234)
    if T <= 1024:
        audio = audio[:, :T].contiguous()
    else:
        audio = torch.cat([audio, audio[:, :T - 1024]], 1).contiguous()
    audio_len = torch.full_like(audio_len, T)
    batch = ((audio, audio_len), tgt)
    for _ in range(4): step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15): step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 15
    print(f"T={T:5d}  T'={(T + 1) // 2:4d}  {dt * 1e3:7.3f} ms/step  {dt * 1e9 / (32 * T):7.1f} ns/frame")
