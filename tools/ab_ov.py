#!/usr/bin/env python3
"""Time the short-K GEMMs of the training step (expand forward, project backward-data; 256- and 512-wide blocks) for ONE library
build (timing-only ablation builds included: results are not checked), rotating buffer sets as tools/gemm_yardstick.py does.
usage: python tools/ab_ov.py [--lib build/variants/lib_X.so] [--iters 48]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--iters", type=int, default=48)
args = ap.parse_args()
if args.lib:
    os.environ["VOICE100_LIB"] = os.path.abspath(args.lib)
from voice100_amd import _native as N  # noqa: E402
from tools.gemm_yardstick import timeit, NSETS  # noqa: E402

N.load()
dev = torch.device("cuda")
B, T = 32, 512
from voice100_amd.functional import pitch16 as _pitch16
P = _pitch16(T, B)
out = []
for (C, hid) in ((256, 1024), (512, 2048)):
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)          # noqa: E731
    W1 = (torch.randn(hid, C, device=dev) / C ** 0.5).to(torch.bfloat16)
    W2t = (torch.randn(hid, C, device=dev) / hid ** 0.5).to(torch.bfloat16)
    sets = [dict(xc=bf(B, C, P), xh2=bf(B, hid, P), yh=torch.empty(B, hid, P, device=dev, dtype=torch.bfloat16)) for _ in range(NSETS)]
    ch = [torch.rand(hid, device=dev) + 0.5 for _ in range(2)]
    st_h = torch.empty(N.helper("v100_pw_num_parts", B, T), hid, 2, device=dev)
    G = lambda *a: N.call("v100_pw_gemm_io", *a)                                      # noqa: E731
    f1 = [(lambda s=s: G(W1, s["xc"], None, None, None, None, 0, s["yh"], None, None, None, 1, st_h, B, hid, C, T, 5)) for s in sets]
    f2 = [(lambda s=s: G(W2t, s["xc"], None, None, None, None, 0, s["yh"], ch[0], ch[1], s["xh2"], 4, st_h, B, hid, C, T, 13)) for s in sets]
    a, b = timeit(f1, args.iters), timeit(f2, args.iters)
    a2, b2 = timeit(f1, args.iters), timeit(f2, args.iters)
    out.append(f"C={C}: expand fwd {min(a, a2) * 1e6:6.1f} us  project bwd-data {min(b, b2) * 1e6:6.1f} us")
print(f"{os.path.basename(args.lib) if args.lib else 'default':12s} " + "   ".join(out))
