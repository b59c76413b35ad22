import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voice100_amd import functional as F_
dev = torch.device("cuda")
for (B, T, V, L) in ((3, 700, 29, 300), (2, 1100, 29, 505), (32, 512, 29, 100)):
    g = torch.Generator().manual_seed(B * 100 + T)
    logits = torch.randn(B, T, V, generator=g) * 2
    targets = torch.randint(1, V, (B, L), generator=g)
    in_len = torch.randint(min(2 * L + 1, T), T + 1, (B,), generator=g).to(torch.int32)
    tgt_len = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    res = {}
    for name, dt in (("f64", torch.float64), ("f32cpu", torch.float32)):
        x = logits.to(dt).clone().requires_grad_(True)
        l = F.ctc_loss(F.log_softmax(x.transpose(0, 1), dim=-1), targets, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
        l.backward(); res[name] = (float(l), x.grad.double())
    x = logits.to(dev).requires_grad_(True)
    l = F_.ctc_loss(x, targets.to(dev), in_len.to(dev), tgt_len.to(dev)); l.backward()
    res["hip"] = (float(l), x.grad.cpu().double())
    ref = res["f64"]
    for k in ("f32cpu", "hip"):
        d = (res[k][1] - ref[1]).norm() / ref[1].norm()
        print(f"B{B} T{T} L{L} {k:7s}: loss err {abs(res[k][0]-ref[0])/abs(ref[0]):.2e}  grad rel-L2 err vs f64 {float(d):.2e}")
