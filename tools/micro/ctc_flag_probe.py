#!/usr/bin/env python3
"""Which (T, logits) make the linear-domain CTC pair flag utterances for its log-domain fallback?  Calls the C entry point with its own
workspace and reads the flag words at the workspace's tail (layout: csrc/ctc.hip ctc_run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice100_amd import _native as N

N.load()
dev = torch.device("cuda:0")
B, V, L = 32, 29, 100


def run(T, scale, seed, blank_bias=0.0):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g) * scale
    logits[:, :, 0] += blank_bias                               # a blank-collapsed model: p(label) ~ e^-bias / 28
    logits = logits.to(dev)
    tgt = torch.randint(1, V, (B, L), generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    tl = torch.full((B,), L, dtype=torch.int32, device=dev)
    nws = N.helper("v100_ctc_workspace_floats", B, T, L)
    ws = torch.zeros(nws, device=dev)
    nll = torch.empty(B, device=dev); loss = torch.empty(1, device=dev); grad = torch.empty_like(logits)
    N.call("v100_ctc_loss_mean", logits, tgt, il, tl, ws, nll, loss, grad, B, T, V, L, 0)
    torch.cuda.synchronize()
    ws_log = 2 * B * T * (2 * L + 1) + B * T + B
    base = (ws.data_ptr() + 4 * ws_log + 7) & ~7              # ll2d [B] doubles
    lat = (base + 8 * B + 15) & ~15                            # lattice rows: 2 * B * T rows of 320 4-byte words
    off = (lat - ws.data_ptr()) // 4 + 2 * B * T * 320
    bad = ws[off:off + B].view(torch.int32)
    return int(bad.sum()), float(loss)


for scale, bias in ((0.05, 0.0), (1.0, 0.0), (2.0, 0.0), (1.0, 6.0), (1.0, 10.0), (2.0, 14.0), (1.0, 20.0)):
    out = []
    for T in (256, 379, 384, 511, 512, 513, 563, 640, 763):
        nb, loss = run(T, scale, T, bias)
        out.append(f"{T}:{nb}")
    print(f"scale {scale} blank bias {bias}: flagged utterances per T  " + " ".join(out) + f"   (loss at the last T {loss:.4f})")
