import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voice100_amd import functional as F_
from voice100_amd.layers import InvertedResidual
F_.set_matmul_precision("bf16")
dev = torch.device("cuda")
torch.manual_seed(3)
cin, k = int(sys.argv[2]), int(sys.argv[3])
blk = InvertedResidual(cin, cin, kernel_size=k).to(dev).train()
x = torch.randn(32, cin, 512, generator=torch.Generator().manual_seed(5)).to(dev).requires_grad_(True)
y = blk(x)
w = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(dev)
(y * w).sum().backward()
out = {"y": y.detach().cpu(), "dx": x.grad.cpu()}
for n, p in blk.named_parameters():
    out["g_" + n] = p.grad.cpu()
for n, b in blk.named_buffers():
    out["b_" + n] = b.detach().cpu().float()
torch.save(out, sys.argv[1])
