#!/usr/bin/env python3
"""Depthwise kernels (bf16 storage, the benchmark's shapes) timed by their own dispatch timestamps in the cache regimes that
matter, to explain the stand-alone -> in-step gap of the graded kernel (round-2 review, item 4a):

  resident   one set of buffers re-used every launch: input + output (134 MB on the 2048-channel layers) stay in the 256 MB
             Infinity Cache -- what tools/bench_kernels.py measures;
  rotating   --sets buffer sets used round-robin (>= 1 GB in flight between two uses of a line): every launch streams from and
             to HBM -- the regime of a training step, where ~2 GB of other tensors pass between two uses;
  produced   rotating, and the input is written by the expand GEMM (v100_pw_gemm_io) launched right before -- the step's own
             order: the depthwise input was just stored (dirty lines in L2 / MALL), its BatchNorm statistics finalised between.

--cm (round 4, review item 3): the same launches on CHANNEL-MAJOR tensors [C][B][P] (io16 bit 16) -- the layout the depthwise access
pattern wants (profiles/r03_stream_pattern_probe.txt) -- with the producing GEMM run as ONE GEMM over all B P columns (B = 1, T = B P).

python tools/bench_dw_regimes.py [--iters 40] [--sets 10] [--lib other.so] [--bwd] [--cm]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voice100_amd import _native as N

LAYERS = [(256, 1024, 19), (256, 1024, 27), (256, 1024, 35), (256, 1024, 51), (512, 2048, 59), (512, 2048, 67), (512, 2048, 75), (512, 2048, 83)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--sets", type=int, default=10)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--bwd", action="store_true", help="also the fused backward kernel")
    ap.add_argument("--da1", action="store_true", help="... and its finished-gradient form (v100_dwconv_bwd_da1_io: what the 16-bit training step runs)")
    ap.add_argument("--only", default="", help="comma list of the columns to run (resident,rotating,produced,bwd resident,bwd rotating,da1 rotating)")
    ap.add_argument("--layers", default="", help="comma list of kernel sizes (default all)")
    ap.add_argument("--cm", action="store_true", help="channel-major tensors [C][B][P]")
    args = ap.parse_args()
    if args.lib:
        N.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda:0")
    B, T = args.B, args.T
    from voice100_amd.functional import pitch16 as _pitch16
    P = _pitch16(T, B)
    want = {int(k) for k in args.layers.split(",") if k}
    tot = {}
    for cin, hid, k in LAYERS:
        if want and k not in want:
            continue
        S = args.sets
        xs = [torch.randn(B, hid, P, device=dev).to(torch.bfloat16) for _ in range(S)]
        ys = [torch.empty(B, hid, P, device=dev, dtype=torch.bfloat16) for _ in range(S)]
        x_in = [torch.randn(B, cin, P, device=dev).to(torch.bfloat16) for _ in range(S)]
        w1 = (torch.randn(hid, cin, device=dev) / cin ** 0.5).to(torch.bfloat16)
        w = torch.randn(hid, k, device=dev) * 0.1
        a, b, c = (torch.randn(hid, device=dev) for _ in range(3))
        G = N.helper("v100_dw_num_groups", B, hid)
        st = torch.empty(max(G, N.helper("v100_pw_num_parts", B, T)), hid, 2, device=dev)
        part = torch.empty(G, hid, k, device=dev)
        dwg = torch.empty(hid, k, device=dev)
        nb_f, nb_b = 2 * B * hid * 2 * T, 2 * B * hid * 4 * T

        cmb = 16 if args.cm else 0                      # (the buffers are just bytes: the same allocations serve both layouts)

        def fwd(i):
            N.call("v100_dwconv_fwd_train_io", xs[i], w, a, b, ys[i], st, G, B, hid, T, k, 9 | cmb)

        def bwd(i):
            j = (i + S // 2) % S
            N.call("v100_dwconv_bwd_io", ys[i], ys[j], w, a, b, c, xs[i], a, b, xs[j], st, part, dwg, G, B, hid, T, k, 15 | cmb)

        pqr = torch.empty(3, hid, device=dev)
        dga, dbe = torch.empty(hid, device=dev), torch.empty(hid, device=dev)
        rstd = torch.rand(hid, device=dev) + 0.5

        def bwd_da1(i):
            j = (i + S // 2) % S
            N.call("v100_dwconv_bwd_da1_io", ys[i], ys[j], w, a, b, c, xs[i], a, b, xs[j], st, dwg, a, b, rstd, pqr, dga, dbe, B, hid, T, k)

        def produce(i):
            if args.cm:
                N.call("v100_pw_gemm_io", w1, x_in[i], None, None, None, None, 0, xs[i], None, None, None, 1, st, 1, hid, cin, B * P, 1 | 4)
            else:
                N.call("v100_pw_gemm_io", w1, x_in[i], None, None, None, None, 0, xs[i], None, None, None, 1, st, B, hid, cin, T, 1 | 4)

        def timed(tag, body):
            for i in range(3):
                body(i % S)
            torch.cuda.synchronize()
            N.timing_enable([tag])
            for i in range(args.iters):
                body(i % S)
            n, ms, _ = N.timing_read()[tag]
            N.timing_enable(False)
            return ms / n * 1e3

        only = {x for x in args.only.split(",") if x}
        cols = [("resident", "dw_fwd", lambda i: fwd(0), nb_f), ("rotating", "dw_fwd", fwd, nb_f),
                ("produced", "dw_fwd", lambda i: (produce(i), fwd(i)), nb_f)]
        if args.bwd:
            cols += [("bwd resident", "dw_bwd_data", lambda i: bwd(0), nb_b), ("bwd rotating", "dw_bwd_data", bwd, nb_b)]
        if args.da1:
            cols += [("da1 rotating", "dw_bwd_data", bwd_da1, nb_b)]
        rows = [(name, timed(tag, body), nb) for name, tag, body, nb in cols if not only or name in only]
        line = f"C={hid:5d} k={k:3d}:"
        for name, us, nb in rows:
            line += f"  {name} {us:6.1f} us {nb / us / 8e6 * 100:5.1f}%"
            t = tot.setdefault(name, [0.0, 0.0])
            t[0] += us; t[1] += nb
        print(line, flush=True)
        del xs, ys, x_in
    for name, (us, nb) in tot.items():
        print(f"TOTAL {name:13s}: {us:7.1f} us  {nb / us / 1e3:7.0f} GB/s = {nb / us / 8e6 * 100:5.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
