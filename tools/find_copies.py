#!/usr/bin/env python3
"""Which host call issues the small device copies seen in the inference kernel tables (__amd_rocclr_copyBuffer)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from voice100_amd import functional as F_
from voice100_amd.asr import AudioToTextCTC
dev = torch.device("cuda")
F_.set_matmul_precision("bf16")
m = AudioToTextCTC(64, 512, 29, 512).to(dev).eval()
x = torch.rand(32, 1024, 64, device=dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m(x)
        torch.cuda.synchronize()
evs = [e for e in prof.events() if "copy" in e.name.lower() or "memcpy" in e.name.lower()]
for e in evs[:40]:
    print(e.name[:60], "| dev us", getattr(e, "device_time", None), "| stack:", [s for s in (e.stack or [])][:4])
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60)[:6000])
