#!/usr/bin/env python3
"""Timing of the v2 conv stacks on one GPU, forward+backward and eval forward, next to the same stack
built from stock PyTorch-ROCm modules (MIOpen conv + LayerNorm + GELU: a yardstick on the same box, not the product) and
per-family kernel times from the library's dispatch timestamps.
  asr_en_base encoder  (config/asr_en_base.yaml:16-18): 64 -> 512 (k5 s2) -> 512 (k5), B=32 x 1024 frames
  tts_en_base decoder  (config/tts_en_base.yaml:20-23): 1024 -> 512 (k5) -> ConvT 512 (k5 s2) -> 512 (k5), B=16 x 512
python tools/bench_v2.py [--iters 30]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from voice100_amd import functional as F_
from voice100_amd import _native as N
from voice100_amd.layers_v2 import get_conv_layers

ASR = (64, [[512, False, 5, 2, 2, False], [512, False, 5, 1, 2, False]], 32, 1024)
TTS = (1024, [[512, False, 5, 1, 2, False], [512, True, 5, 2, 2, False], [512, False, 5, 1, 2, False]], 16, 512)


class StockBlock(nn.Module):
    def __init__(self, cin, cout, transpose, k, stride, padding, bias):
        super().__init__()
        conv = nn.ConvTranspose1d if transpose else nn.Conv1d
        self.conv = conv(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=bias)
        self.layer_norm = nn.LayerNorm(cout)

    def forward(self, x):
        x = self.conv(x)
        x = self.layer_norm(x.transpose(-2, -1)).transpose(-2, -1)
        return nn.functional.gelu(x)


def stock(cin, settings):
    layers = []
    for cout, transpose, k, stride, padding, bias in settings:
        layers.append(StockBlock(cin, cout, transpose, k, stride, padding, bias))
        cin = cout
    return nn.Sequential(*layers)


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--profile-case", default=None, help="asr_encoder|tts_decoder: only the HIP bf16 training step (for rocprofv3)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    out = {}
    if args.profile_case:
        cin, settings, B, T = {"asr_encoder": ASR, "tts_decoder": TTS}[args.profile_case]
        x = torch.randn(B, cin, T, device=dev, requires_grad=True)
        F_.set_matmul_precision("bf16")
        m = get_conv_layers(cin, settings).to(dev)
        gy = torch.randn_like(m(x))
        for _ in range(args.iters):
            m(x).backward(gy)
        torch.cuda.synchronize()
        return
    for name, (cin, settings, B, T) in (("asr_encoder", ASR), ("tts_decoder", TTS)):
        x = torch.randn(B, cin, T, device=dev, requires_grad=True)
        res = {}
        ref = stock(cin, settings).to(dev)

        def train_step(m):
            y = m(x)
            y.backward(gy)
            return y

        with torch.no_grad():
            gy = torch.randn_like(ref(x))
        res["stock_torch_fp32_train_ms"] = round(timeit(lambda: train_step(ref), args.iters), 3)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            res["stock_torch_autocast_bf16_train_ms"] = round(timeit(lambda: train_step(ref), args.iters), 3)
        with torch.no_grad():
            res["stock_torch_fp32_eval_ms"] = round(timeit(lambda: ref(x), args.iters), 3)
        for prec in ("fp32", "bf16"):
            F_.set_matmul_precision(prec)
            m = get_conv_layers(cin, settings).to(dev)
            res[f"hip_{prec}_train_ms"] = round(timeit(lambda: train_step(m), args.iters), 3)
            N.timing_enable(True)
            for _ in range(3):
                train_step(m)
            torch.cuda.synchronize()
            res[f"hip_{prec}_train_kernel_ms"] = {k: round(v[1] / 3, 3) for k, v in sorted(N.timing_read().items())}
            N.timing_enable(False)
            with torch.no_grad():
                res[f"hip_{prec}_eval_ms"] = round(timeit(lambda: m(x), args.iters), 3)
        F_.set_matmul_precision("fp32")
        flops = 0
        t, c = T, cin
        for cout, transpose, k, stride, padding, bias in settings:
            t = (t - 1) * stride - 2 * padding + k if transpose else (t + 2 * padding - k) // stride + 1
            macs = B * cout * c * k * (t if not transpose else (t + 1) // 2)
            flops += 2 * macs
            c = cout
        res["fwd_gflop"] = round(flops / 1e9, 1)
        res["hip_bf16_train_tflops"] = round(3 * flops / (res["hip_bf16_train_ms"] * 1e-3) / 1e12, 1)
        out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
