import os, subprocess, sys, torch
root = sys.argv[1]
script = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from voice100_amd import functional as F_
from voice100_amd.layers import InvertedResidual
F_.set_matmul_precision("bf16")
torch.manual_seed(5)
dev = torch.device("cuda:0")
net = torch.nn.Sequential(InvertedResidual(256, 256, kernel_size=19)).to(dev).train()
out = {}
for T in (200, 512):
    g = torch.Generator().manual_seed(T)
    x = torch.randn(8, 256, T, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(8, 256, T, generator=g).to(dev)
    for p in net.parameters():
        p.grad = None
    y = F_.ir_stack_train(list(net), x)
    (y * gy).sum().backward()
    out[T] = [y.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() for p in net.parameters()]
torch.save(out, sys.argv[2])
"""
open("/tmp/da1s.py", "w").write(script)
res = {}
for v in ("1", "0"):
    subprocess.run([sys.executable, "/tmp/da1s.py", root, f"/tmp/o{v}.pt"], check=True, env=dict(os.environ, V100_IR_DA1=v))
    res[v] = torch.load(f"/tmp/o{v}.pt")
names = ["y", "dx", "w1", "g1", "b1", "wd", "g2", "b2", "w3", "g3", "b3"]
for T in res["1"]:
    for n, a, b in zip(names, res["1"][T], res["0"][T]):
        d = (a - b).abs().max().item()
        print(T, n, tuple(a.shape), "maxdiff", d, "max", b.abs().max().item(), "nan", bool(torch.isnan(a).any()))
