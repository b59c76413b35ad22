#!/usr/bin/env python3
"""The data-parallel machinery on ONE rank (one-rank RCCL group, force_exchange, three-block segments) against the plain step, same
process, interleaved windows: where do the extra ~0.3 ms of `dp_path_single_rank` go?  Run plain or under rocprofv3 --kernel-trace --stats.
    python tools/prof_dp.py [--steps 30] [--mode dp|plain|both]"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench
from voice100_amd import functional as F_
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--mode", default="both")
    ap.add_argument("--no-timestretch", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    F_.set_matmul_precision("bf16")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
    if args.no_timestretch:
        model.batch_augment.do_timestretch = False
    batch = bench.synth_batch(dev, 32, 1234)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1)

    def window(step, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(batch)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3

    res = {}
    if args.mode in ("plain", "both"):
        F_.set_stack_segment(None)
        plain = TrainStep(model)
        for _ in range(5):
            plain(batch)
        res["plain"] = [window(plain, args.steps) for _ in range(3)]
        plain.buckets.remove_hooks()
    if args.mode in ("dp", "both"):
        F_.set_stack_segment(3)
        dp = TrainStep(model, force_exchange=True)
        for _ in range(5):
            dp(batch)
        res["dp"] = [window(dp, args.steps) for _ in range(3)]
        dp.buckets.remove_hooks()
        F_.set_stack_segment(1)
        # segments only (no exchange): what the three autograd nodes cost
        F_.set_stack_segment(3)
        seg = TrainStep(model)
        for _ in range(5):
            seg(batch)
        res["segments_only"] = [window(seg, args.steps) for _ in range(3)]
    for k, v in res.items():
        print(k, " ".join(f"{a:.3f}/{b:.3f}" for a, b in v), "(ms per step / host enqueue ms per step, 3 windows)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
