#!/usr/bin/env python3
"""Micro-benchmark of the act16 1x1 GEMMs (bf16-stored hidden tensors) at the step's two main block shapes.
usage: python tools/bench_gemm_io.py [--lib build/variants/lib_X.so] [--iters 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
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
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=512)
    args = ap.parse_args()
    if args.lib:
        os.environ["VOICE100_LIB"] = os.path.abspath(args.lib)
    from voice100_amd import _native as N
    dev = torch.device("cuda")
    B, T = args.B, args.T
    from voice100_amd.functional import pitch16 as _pitch16
    P = _pitch16(T, B)
    tot = 0.0
    for (C, hid) in ((256, 1024), (512, 2048)):
        bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
        x = torch.randn(B, C, T, device=dev)
        a1, a2, dz1, dz2 = bf(B, hid, P), bf(B, hid, P), bf(B, hid, P), bf(B, hid, P)
        a3, da3 = bf(B, C, P), bf(B, C, P)
        y = torch.empty(B, C, T, device=dev)
        W1 = (torch.randn(hid, C, device=dev) / C ** 0.5).to(torch.bfloat16)
        W2 = (torch.randn(C, hid, device=dev) / hid ** 0.5).to(torch.bfloat16)
        W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
        ch, cc = [torch.randn(hid, device=dev) for _ in range(3)], [torch.randn(C, device=dev) for _ in range(3)]
        parts = N.helper("v100_pw_num_parts", B, T)
        st_h, st_c = torch.empty(parts, hid, 2, device=dev), torch.empty(parts, C, 2, device=dev)
        S1, S2 = N.helper("v100_pw_wgrad_splits", B, hid, C), N.helper("v100_pw_wgrad_splits", B, C, hid)
        p1, p2 = torch.empty(S1, hid, C, device=dev), torch.empty(S2, C, hid, device=dev)
        dW1, dW2 = torch.empty(hid, C, device=dev), torch.empty(C, hid, device=dev)
        fl = 2.0 * B * hid * C * T
        cases = [
            ("expand fwd   <0,1,Y>", lambda: N.call("v100_pw_gemm_io", W1, x, None, None, None, None, 0, a1, None, None, None, 1, st_h, B, hid, C, T, 4),
             B * T * (4 * C + 2 * hid)),
            ("expand fwd   <0,1,X|Y>", lambda: N.call("v100_pw_gemm_io", W1, a3, None, None, None, None, 0, a1, None, None, None, 1, st_h, B, hid, C, T, 5),
             B * T * (2 * C + 2 * hid)),
            ("project fwd  <1,1,X|Y>", lambda: N.call("v100_pw_gemm_io", W2, a2, None, ch[0], ch[1], None, 1, a3, None, None, None, 1, st_c, B, C, hid, T, 5),
             B * T * (2 * hid + 2 * C)),
            ("project bwdd <0,4,X|Y|R>", lambda: N.call("v100_pw_gemm_io", W2t, da3, None, None, None, None, 0, dz2, ch[0], ch[1], a2, 4, st_h, B, hid, C, T, 13),
             B * T * (2 * C + 4 * hid)),
            ("expand bwdd  <2,5,X|X2>", lambda: N.call("v100_pw_gemm_io", W1t, dz1, a1, ch[0], ch[1], ch[2], 2, y, None, None, x, 5, None, B, C, hid, T, 3),
             B * T * (4 * hid + 8 * C)),
            ("expand wgrad <2,0,G|G2>", lambda: N.call("v100_pw_wgrad_io", dz1, a1, ch[0], ch[1], ch[2], 2, x, None, None, 0, p1, dW1, S1, B, hid, C, T, 3),
             B * T * (4 * hid + 4 * C)),
            ("expand wgrad <2,0,G|G2|X>", lambda: N.call("v100_pw_wgrad_io", dz1, a1, ch[0], ch[1], ch[2], 2, a3, None, None, 0, p1, dW1, S1, B, hid, C, T, 7),
             B * T * (4 * hid + 2 * C)),
            ("project wgrad<0,1,G|X>", lambda: N.call("v100_pw_wgrad_io", da3, None, None, None, None, 0, a2, ch[0], ch[1], 1, p2, dW2, S2, B, C, hid, T, 5),
             B * T * (2 * hid + 2 * C)),
        ]
        for name, fn, nbytes in cases:
            dt = timeit(fn, args.iters)
            tot += dt
            print(f"C={C:4d} hid={hid:5d} {name:26s}: {dt*1e6:7.1f} us  {fl/dt/1e12:6.0f} TFLOP/s  {nbytes/dt/1e9:6.0f} GB/s (floor {nbytes/4.7e12*1e6:5.1f} us hbm, {fl/2.5e15*1e6:4.1f} us mfma)")
    print(f"TOTAL {tot*1e6:.1f} us")


if __name__ == "__main__":
    main()
