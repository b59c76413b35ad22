#!/usr/bin/env python3
"""The bench's own model and batch: after a few training steps, which utterances does the linear-domain CTC pair flag on the real logits,
and how do its loss / gradient compare with torch float64?"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch, torch.nn.functional as F
import bench
from voice100_amd import functional as F_, _native as N
from voice100_amd.asr import AudioToTextCTC
from voice100_amd.trainer import TrainStep

dev = torch.device("cuda:0")
N.load()
F_.set_matmul_precision("bf16")
random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
model = AudioToTextCTC(64, 512, 29, 512, learning_rate=1e-3, weight_decay=4e-5).to(dev)
step = TrainStep(model)
batch = bench.synth_batch(dev, 32, 1234)
for nsteps in (0, 5, 20, 40):
    while step.step_idx < nsteps:
        step(batch)
    model.eval()
    with torch.no_grad():
        logits = model(batch[0][0]).float().contiguous()
    model.train()
    B, T, V = logits.shape
    (text, text_len) = batch[1]
    L = text.shape[1]
    il = torch.full((B,), T, dtype=torch.int32, device=dev)
    nws = N.helper("v100_ctc_workspace_floats", B, T, L)
    ws = torch.zeros(nws, device=dev)
    nll = torch.empty(B, device=dev); loss = torch.empty(1, device=dev); grad = torch.empty_like(logits)
    N.call("v100_ctc_loss_mean", logits, text, il, text_len, ws, nll, loss, grad, B, T, V, L, 0)
    torch.cuda.synchronize()
    ws_log = 2 * B * T * (2 * L + 1) + B * T + B
    base = (ws.data_ptr() + 4 * ws_log + 7) & ~7
    lat = (base + 8 * B + 15) & ~15
    off = (lat - ws.data_ptr()) // 4 + 2 * B * T * 320
    bad = ws[off:off + B].view(torch.int32).cpu()
    ref_in = logits.double().cpu().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(ref_in.transpose(0, 1), dim=-1), text.cpu(), il.cpu(), text_len.cpu(), blank=0, reduction="mean", zero_infinity=True)
    ref.backward()
    gr = ref_in.grad.float()
    err = float((grad.cpu() - gr).abs().max() / gr.abs().max())
    lp = F.log_softmax(logits, -1)
    print(f"after {nsteps} steps: flagged {int(bad.sum())} of {B}; loss {float(loss):.6f} ref {float(ref):.6f}; grad max-rel-err {err:.2e}; "
          f"logit range [{float(logits.min()):.2f}, {float(logits.max()):.2f}], mean blank log-prob {float(lp[..., 0].mean()):.3f}, min log-prob {float(lp.min()):.2f}")
