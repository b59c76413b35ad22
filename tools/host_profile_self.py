#!/usr/bin/env python3
"""cProfile of the training step with autograd's worker thread switched off (torch.autograd.set_multithreading_enabled(False)), so that the
backward's Python side -- the stack executor's call, the per-block functions -- shows up in the profile instead of one opaque run_backward."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from voice100_amd import functional as F_, _native as N
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep
dev = torch.device("cuda:0")
N.load(); F_.set_matmul_precision("bf16")
torch.manual_seed(1234)
model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
step = TrainStep(model)
batch = bench.synth_batch(dev, 32, 1234)
torch.autograd.set_multithreading_enabled(False)
for _ in range(5): step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(40): step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
