import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voice100_amd import _native as N
dev = torch.device("cuda")
for (B, M, K, T) in ((1, 512, 128, 700), (2, 256, 64, 256), (1, 512, 128, 640)):
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev)
    Abf = A.to(torch.bfloat16)
    x = torch.randn(B, K, T, generator=g).to(dev)
    parts = N.helper("v100_pw_num_parts", B, T)
    y = torch.full((B, M, T), float("nan"), device=dev)
    st = torch.zeros(parts, M, 2, device=dev)
    N.call("v100_pw_gemm", A, Abf, x, None, None, None, None, 0, y, None, None, None, None, 1, st, B, M, K, T, 1)
    ref = torch.matmul(Abf.float(), x.to(torch.bfloat16).float())
    d = (y - ref).abs()
    bad = (d > 1e-3) | torch.isnan(y)
    print(B, M, K, T, "max err", float(d[~torch.isnan(y)].max()), "bad", int(bad.sum()), "nan", int(torch.isnan(y).sum()))
    if bad.any():
        idx = bad.nonzero()
        print(" first bad", idx[:5].tolist(), " last bad", idx[-3:].tolist())
        print(" bad rows (m) unique count", idx[:, 1].unique().numel(), "bad cols min/max", int(idx[:, 2].min()), int(idx[:, 2].max()))
    P = (T + 7) & ~7
    y16 = torch.full((B, M, P), float("nan"), dtype=torch.bfloat16, device=dev)
    N.call("v100_pw_gemm_io", Abf, x, None, None, None, None, 0, y16, None, None, None, 1, st, B, M, K, T, 4)
    d = (y16[:, :, :T].float() - ref).abs()
    bad = (d > 3e-2) | torch.isnan(y16[:, :, :T].float())
    print("   io: bad", int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print(" first bad", idx[:5].tolist(), " last bad", idx[-3:].tolist())
        print(" bad rows unique", idx[:, 1].unique().numel(), "cols min/max", int(idx[:, 2].min()), int(idx[:, 2].max()))
