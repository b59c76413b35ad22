#!/usr/bin/env python3
"""Linear-domain CTC pair against torch's float64 CPU CTC on bench-like inputs (B = 32, T' = 512, V = 29, 100-token targets):
loss and gradient errors for flat / peaky / blank-biased logits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from voice100_amd import functional as F_

dev = torch.device("cuda:0")
B, V, L = 32, 29, 100
for T, scale, bias in ((512, 0.05, 0.0), (512, 0.3, 0.0), (512, 1.0, 0.0), (512, 0.3, 3.0), (512, 1.0, 6.0), (763, 0.1, 1.0), (379, 0.1, 0.5)):
    g = torch.Generator().manual_seed(T + int(scale * 100))
    logits = torch.randn(B, T, V, generator=g) * scale
    logits[:, :, 0] += bias
    tgt = torch.randint(1, V, (B, L), generator=g)
    il = torch.full((B,), T, dtype=torch.int32); tl = torch.full((B,), L, dtype=torch.int32)
    ref_in = logits.double().clone().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(ref_in.transpose(0, 1), dim=-1), tgt, il, tl, blank=0, reduction="mean", zero_infinity=True)
    ref.backward()
    x = logits.to(dev).requires_grad_(True)
    loss = F_.ctc_loss(x, tgt.to(dev), il.to(dev), tl.to(dev))
    loss.backward()
    gr = ref_in.grad.float()
    err = float((x.grad.cpu() - gr).abs().max() / gr.abs().max())
    print(f"T {T} scale {scale} bias {bias}: loss {float(loss):.6f} ref {float(ref):.6f}  grad max-rel-err {err:.2e}")
