#!/usr/bin/env python3
"""Time the CTC head alone at the bench's size (B = 32, T' = 512, V = 29, 100-token targets): lse + lattice + gradient kernels.
VOICE100_LIB selects the library (A/B variants from tools/ab_variants.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice100_amd import functional as F_

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, T, V, L = 32, 512, 29, int(os.environ.get("CTC_L", "100"))
logits = torch.randn(B, T, V, generator=g).to(dev)
tgt = torch.randint(1, V, (B, L), generator=g).to(dev)
il = torch.randint(T * 3 // 4, T + 1, (B,), generator=g).to(torch.int32).to(dev)
tl = torch.full((B,), L, dtype=torch.int32, device=dev)
for _ in range(5):
    loss = F_.ctc_loss(logits, tgt, il, tl)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
a.record()
for _ in range(n):
    loss = F_.ctc_loss(logits, tgt, il, tl)
b.record()
torch.cuda.synchronize()
print(f"{os.environ.get('VOICE100_LIB', 'default')}: ctc_loss {a.elapsed_time(b) / n * 1e3:.1f} us/call  loss {float(loss):.6f}")
