#!/usr/bin/env python3
"""Per-layer micro-benchmark of the hot kernels at the metric's shapes (B=32, T=1024):
depthwise fwd / bwd-data / wgrad per encoder layer (GB/s of algorithmic bytes, % of 8 TB/s) and the
1x1 GEMMs (TFLOP/s).  GPU only.  python tools/bench_kernels.py [--iters 20] [--what dw,pw]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voice100_amd import _native as N

SPECS = [(64, 256, 256, 11, 2), (256, 1024, 256, 19, 1), (256, 1024, 256, 27, 1), (256, 1024, 256, 35, 1), (256, 1024, 512, 51, 1),
         (512, 2048, 512, 59, 1), (512, 2048, 512, 67, 1), (512, 2048, 512, 75, 1), (512, 2048, 512, 83, 1)]


def timeit(fn, iters):
    for _ in range(max(3, iters // 4)):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--what", default="dw,pw")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--lib", default=None, help="alternative libvoice100_hip.so (A/B builds of one kernel in one run)")
    args = ap.parse_args()
    if args.lib:
        N.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda:0")
    B, T = args.B, args.T
    tot = {"dw_fwd": [0, 0.0], "dw_bwd": [0, 0.0], "dw_wgrad": [0, 0.0], "dw_bwd_fused": [0, 0.0], "dw16_fwd": [0, 0.0], "dw16_bwd_fused": [0, 0.0]}
    t = T
    for (cin, hid, cout, k, s) in SPECS:
        tout = (t - 1) // s + 1
        pad = (k - 1) // 2
        if "dw" in args.what.split(","):
            x = torch.randn(B, hid, t, device=dev)
            w = torch.randn(hid, k, device=dev) * 0.1
            a, b, c = (torch.randn(hid, device=dev) for _ in range(3))
            y = torch.empty(B, hid, tout, device=dev)
            y2 = torch.randn(B, hid, tout, device=dev)
            G = N.helper("v100_dw_num_groups", B, hid)
            st = torch.empty(G, hid, 2, device=dev)
            dz1 = torch.empty(B, hid, t, device=dev)
            part = torch.empty(G, hid, k, device=dev)
            dw = torch.empty(hid, k, device=dev)
            f = lambda: N.call("v100_dwconv", x, None, w, a, b, None, 1, y, None, None, None, 0, st, G, B, hid, t, tout, k, s, pad, 0, 1, 0)
            bd = lambda: N.call("v100_dwconv", y, y2, w, a, b, c, 2, dz1, x, a, b, 2, st, G, B, hid, tout, t, k, 1, k - 1 - pad, 1, s, 0)
            wg = lambda: N.call("v100_dwconv_wgrad", y, y2, a, b, c, 2, x, a, b, 1, part, dw, G, B, hid, t, tout, k, s, pad, 0)
            bf_ = lambda: N.call("v100_dwconv_bwd", y, y2, w, a, b, c, x, a, b, dz1, st, part, dw, G, B, hid, t, tout, k, s, pad, 0)
            bytes_f = 4 * B * hid * (t + tout)
            bytes_b = 4 * B * hid * (2 * tout + 2 * t)
            bytes_w = 4 * B * hid * (2 * tout + t)
            for name, fn, nb in (("dw_fwd", f, bytes_f), ("dw_bwd", bd, bytes_b), ("dw_wgrad", wg, bytes_w), ("dw_bwd_fused", bf_, bytes_b)):
                dt = timeit(fn, args.iters)
                tot[name][0] += nb; tot[name][1] += dt
                fl = 2 * B * hid * k * tout
                print(f"{name:12s} C={hid:5d} k={k:3d} s={s} T={t:5d}: {dt*1e6:8.1f} us  {nb/dt/1e9:7.0f} GB/s ({nb/dt/8e12*100:5.1f}% of 8TB/s)  {fl/dt/1e12:6.1f} TFLOP/s")
        if "dw16" in args.what.split(",") and s == 1:
            # bf16 storage of the hidden tensors (act16): same layers, bf16 [B, C, pitch] in / out
            from voice100_amd.functional import pitch16 as _pitch16
            P = _pitch16(t, B)
            x16 = torch.randn(B, hid, P, device=dev).to(torch.bfloat16)
            y16 = torch.empty(B, hid, P, device=dev, dtype=torch.bfloat16)
            g16 = torch.randn(B, hid, P, device=dev).to(torch.bfloat16)
            a216 = torch.randn(B, hid, P, device=dev).to(torch.bfloat16)
            dz116 = torch.empty(B, hid, P, device=dev, dtype=torch.bfloat16)
            w = torch.randn(hid, k, device=dev) * 0.1
            a, b, c = (torch.randn(hid, device=dev) for _ in range(3))
            G = N.helper("v100_dw_num_groups", B, hid)
            st = torch.empty(G, hid, 2, device=dev)
            part = torch.empty(G, hid, k, device=dev)
            dw = torch.empty(hid, k, device=dev)
            f16 = lambda: N.call("v100_dwconv_fwd_train_io", x16, w, a, b, y16, st, G, B, hid, t, k, 9)
            b16 = lambda: N.call("v100_dwconv_bwd_io", g16, a216, w, a, b, c, x16, a, b, dz116, st, part, dw, G, B, hid, t, k, 15)
            for name, fn, nb in (("dw16_fwd", f16, 2 * B * hid * 2 * t), ("dw16_bwd_fused", b16, 2 * B * hid * 4 * t)):
                dt = timeit(fn, args.iters)
                tot[name][0] += nb; tot[name][1] += dt
                print(f"{name:14s} C={hid:5d} k={k:3d} T={t:5d}: {dt*1e6:8.1f} us  {nb/dt/1e9:7.0f} GB/s ({nb/dt/8e12*100:5.1f}% of 8TB/s)")
        if "pw" in args.what.split(","):
            for bf in (0, 1):
                for (M, K, TT, tag) in ((hid, cin, t, "pw1"), (cout, hid, tout, "pw2")):
                    X = torch.randn(B, K, TT, device=dev)
                    A = torch.randn(M, K, device=dev) / K ** 0.5
                    Abf = A.to(torch.bfloat16)
                    Y = torch.empty(B, M, TT, device=dev)
                    parts = N.helper("v100_pw_num_parts", B, TT)
                    st = torch.empty(parts, M, 2, device=dev)
                    xa, xb = torch.randn(K, device=dev), torch.randn(K, device=dev)
                    fn = lambda: N.call("v100_pw_gemm", A, Abf, X, None, xa, xb, None, 1, Y, None, None, None, None, 1, st, B, M, K, TT, bf)
                    dt = timeit(fn, args.iters)
                    fl = 2.0 * B * M * K * TT
                    byts = 4 * B * TT * (M + K)
                    print(f"{tag} {'bf16' if bf else 'f32 '} M={M:5d} K={K:5d} T={TT:5d}: {dt*1e6:8.1f} us  {fl/dt/1e12:7.1f} TFLOP/s  {byts/dt/1e9:7.0f} GB/s")
                    S = N.helper("v100_pw_wgrad_splits", B, M, K)
                    partial = torch.empty(S, M, K, device=dev)
                    dW = torch.empty(M, K, device=dev)
                    fn = lambda: N.call("v100_pw_wgrad", Y, None, None, None, None, 0, X, xa, xb, 1, partial, dW, S, B, M, K, TT, bf)
                    dt = timeit(fn, args.iters)
                    print(f"{tag} wgrad {'bf16' if bf else 'f32 '} M={M:5d} K={K:5d} T={TT:5d} S={S}: {dt*1e6:8.1f} us  {fl/dt/1e12:7.1f} TFLOP/s")
        t = tout
    for name, (nb, dt) in tot.items():
        if dt:
            print(f"TOTAL {name}: {dt*1e6:.1f} us, {nb/1e9:.3f} GB -> {nb/dt/1e9:.0f} GB/s = {nb/dt/8e12*100:.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
