#!/usr/bin/env python3
"""Yardstick for the 1x1-GEMM family (review round 3, item 1a): what a tuned PLAIN bf16 GEMM does on the six shapes of a 256-wide
and a 512-wide InvertedResidual block on this chip, next to this library's kernels on the same shapes.

  yardstick : torch.matmul on bf16 tensors (hipBLASLt / rocBLAS underneath) -- a MEASUREMENT ONLY, never on the product path --
              (a) "batched": the library's own [B][C][T] layout, B GEMMs per call (torch broadcasts the weight),
              (b) "flat"   : one GEMM over all B*T columns of a channel-major [C][B*T] operand (what a channel-major layout would give);
              the two weight gradients contract over B*T (flat: one GEMM; batched: bmm + sum is not what anyone would run, so flat only).
  ours      : v100_pw_gemm_io / v100_pw_wgrad_io with (i) plain operands and the plain store epilogue (transforms and statistics
              ablated: the comparable configuration), (ii) the modes the training step really runs (transform on load, statistics /
              mask epilogues, bf16-stored hidden tensors).
Every timing rotates over NSETS buffer sets (> 256 MB between two uses of a line: the Infinity Cache does not serve the operands).

usage (GPU box):  python tools/gemm_yardstick.py [--iters 40] > profiles/r04_gemm_yardstick.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

NSETS = 6


def timeit(fns, iters):
    """fns: list of NSETS callables (one per buffer set), called round-robin; returns seconds per call."""
    n = len(fns)
    for i in range(2 * n):
        fns[i % n]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fns[i % n]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=512)
    args = ap.parse_args()
    from voice100_amd import _native as N
    N.load()
    dev = torch.device("cuda")
    B, T = args.B, args.T
    from voice100_amd.functional import pitch16 as _pitch16
    P = _pitch16(T, B)
    Ncol = B * T
    print(f"# device: {torch.cuda.get_device_name(0)}; torch {torch.__version__}; B = {B}, T = {T} (N = B*T = {Ncol} columns); {NSETS} rotating buffer sets")
    print(f"# peak used for the fractions: 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)")
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)          # noqa: E731
    for (C, hid) in ((256, 1024), (512, 2048)):
        fl = 2.0 * hid * C * Ncol
        W1 = (torch.randn(hid, C, device=dev) / C ** 0.5).to(torch.bfloat16)        # expand weight  [hid][C]
        W2 = (torch.randn(C, hid, device=dev) / hid ** 0.5).to(torch.bfloat16)      # project weight [C][hid]
        W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
        sets = []
        for _ in range(NSETS):
            s = dict(xc=bf(B, C, P), xh=bf(B, hid, P), xh2=bf(B, hid, P), yc=torch.empty(B, C, P, device=dev, dtype=torch.bfloat16),
                     yh=torch.empty(B, hid, P, device=dev, dtype=torch.bfloat16), yc32=torch.empty(B, C, T, device=dev),
                     x32=torch.randn(B, C, T, device=dev))
            s["fc"] = s["xc"].permute(1, 0, 2).reshape(C, Ncol).contiguous()       # channel-major copies for the flat yardstick
            s["fh"] = s["xh"].permute(1, 0, 2).reshape(hid, Ncol).contiguous()
            s["ofc"] = torch.empty(C, Ncol, device=dev, dtype=torch.bfloat16)
            s["ofh"] = torch.empty(hid, Ncol, device=dev, dtype=torch.bfloat16)
            sets.append(s)
        ch = [torch.rand(hid, device=dev) + 0.5 for _ in range(3)]
        cc = [torch.rand(C, device=dev) + 0.5 for _ in range(2)]
        parts = N.helper("v100_pw_num_parts", B, T)
        st_h, st_c = torch.empty(parts, hid, 2, device=dev), torch.empty(parts, C, 2, device=dev)
        S1, S2 = N.helper("v100_pw_wgrad_splits", B, hid, C), N.helper("v100_pw_wgrad_splits", B, C, hid)
        p1, p2 = torch.empty(S1, hid, C, device=dev), torch.empty(S2, C, hid, device=dev)
        dW1, dW2 = torch.empty(hid, C, device=dev), torch.empty(C, hid, device=dev)
        dW1b, dW2b = torch.empty(hid, C, device=dev, dtype=torch.bfloat16), torch.empty(C, hid, device=dev, dtype=torch.bfloat16)

        def L(f):
            return [(lambda s=s: f(s)) for s in sets]

        G = lambda *a: N.call("v100_pw_gemm_io", *a)                                  # noqa: E731
        WG = lambda *a: N.call("v100_pw_wgrad_io", *a)                                # noqa: E731
        rows = [
            # name, M, K, yardstick batched, yardstick flat, ours plain, ours in-step (mode string)
            ("expand fwd        Y[hid x N] = W1[hid x C] X[C x N]", hid, C,
             L(lambda s: torch.matmul(W1, s["xc"], out=s["yh"])), L(lambda s: torch.matmul(W1, s["fc"], out=s["ofh"])),
             None,
             L(lambda s: G(W1, s["xc"], None, None, None, None, 0, s["yh"], None, None, None, 1, st_h, B, hid, C, T, 5)), "plain X (no transform exists), bf16 out +BN sums"),
            ("project fwd       Y[C x N] = W2[C x hid] X[hid x N]", C, hid,
             L(lambda s: torch.matmul(W2, s["xh"], out=s["yc"])), L(lambda s: torch.matmul(W2, s["fh"], out=s["ofc"])),
             L(lambda s: G(W2, s["xh"], None, None, None, None, 0, s["yc32"], cc[0], cc[1], None, 3, None, B, C, hid, T, 1)),
             L(lambda s: G(W2, s["xh"], None, ch[0], ch[1], None, 1, s["yc"], None, None, None, 1, st_c, B, C, hid, T, 5)), "BN+ReLU6 on load, +BN sums (plain: the eval-mode kernel, fp32 out)"),
            ("project bwd-data  Y[hid x N] = W2^T[hid x C] X[C x N]", hid, C,
             L(lambda s: torch.matmul(W2t, s["xc"], out=s["yh"])), L(lambda s: torch.matmul(W2t, s["fc"], out=s["ofh"])),
             L(lambda s: G(W2t, s["xc"], None, None, None, None, 0, s["yh"], None, None, None, 1, st_h, B, hid, C, T, 5)),
             L(lambda s: G(W2t, s["xc"], None, None, None, None, 0, s["yh"], ch[0], ch[1], s["xh2"], 4, st_h, B, hid, C, T, 13)), "plain X, ReLU6 mask + BN-bwd sums (plain: the +BN-sums epilogue)"),
            ("expand bwd-data   Y[C x N] = W1^T[C x hid] X[hid x N]", C, hid,
             L(lambda s: torch.matmul(W1t, s["xh"], out=s["yc"])), L(lambda s: torch.matmul(W1t, s["fh"], out=s["ofc"])),
             L(lambda s: G(W1t, s["xh"], None, None, None, None, 0, s["yc32"], cc[0], cc[1], None, 3, None, B, C, hid, T, 1)),
             L(lambda s: G(W1t, s["xh"], None, None, None, None, 0, s["yc32"], None, None, s["x32"], 5, None, B, C, hid, T, 1)), "round 5: plain bf16 X = the finished gradient da1 (written by the depthwise backward), +residual, fp32 out"),
            ("  ... round 4's form: BN-bwd affine of (dz1, a1) on load", C, hid, None, None, None,
             L(lambda s: G(W1t, s["xh"], s["xh2"], ch[0], ch[1], ch[2], 2, s["yc32"], None, None, s["x32"], 5, None, B, C, hid, T, 3)), "BN-bwd affine of two tensors on load, +residual, fp32 out"),
            ("expand wgrad      dW1[hid x C] = G[hid x N] X[C x N]^T", hid, C,
             None, L(lambda s: torch.matmul(s["fh"], s["fc"].t(), out=dW1b)),
             None,
             L(lambda s: WG(s["xh"], None, None, None, None, 0, s["xc"], None, None, 0, p1, dW1, S1, B, hid, C, T, 5)), "round 5: plain bf16 G = da1 (+ slab reduce launch)"),
            ("  ... round 4's form: BN-bwd affine on G", hid, C, None, None, None,
             L(lambda s: WG(s["xh"], s["xh2"], ch[0], ch[1], ch[2], 2, s["xc"], None, None, 0, p1, dW1, S1, B, hid, C, T, 7)), "BN-bwd affine on G (+ slab reduce launch)"),
            ("project wgrad     dW2[C x hid] = G[C x N] X[hid x N]^T", C, hid,
             None, L(lambda s: torch.matmul(s["fc"], s["fh"].t(), out=dW2b)),
             None,
             L(lambda s: WG(s["xc"], None, None, None, None, 0, s["xh"], ch[0], ch[1], 1, p2, dW2, S2, B, C, hid, T, 5)), "BN+ReLU6 on X (+ slab reduce launch)"),
        ]
        print(f"\n## block width C = {C}, hidden {hid}: {fl/1e9:.1f} GFLOP per GEMM, matrix-pipe floor {fl/2.5e15*1e6:.1f} us")
        print(f"{'GEMM':58s} {'hipBLASLt batched':>18s} {'hipBLASLt flat':>16s} {'ours plain':>14s} {'ours in-step':>14s}   in-step mode")
        tot = [0.0, 0.0, 0.0, 0.0]
        for name, M, K, yb, yf, op, oi, mode in rows:
            cells = []
            extra = name.startswith("  ...")                 # a second form of the row above: shown, not summed
            for j, fns in enumerate((yb, yf, op, oi)):
                if fns is None:
                    cells.append(f"{'-':>16s}")
                    continue
                try:
                    dt = timeit(fns, args.iters)
                    if not extra:
                        tot[j] += dt
                    cells.append(f"{dt*1e6:7.1f} us {fl/dt/2.5e15:5.2f}")
                except Exception as ex:                                      # noqa: BLE001
                    cells.append(f"{'ERR ' + type(ex).__name__:>16s}")
            print(f"{name:58s} {cells[0]:>18s} {cells[1]:>16s} {cells[2]:>14s} {cells[3]:>14s}   {mode}")
        print(f"{'sum (batched column: 4 GEMMs; others: 6)':58s} {tot[0]*1e6:15.1f} us {tot[1]*1e6:13.1f} us {tot[2]*1e6:11.1f} us {tot[3]*1e6:11.1f} us")
        # Layout A/B inside OUR kernels (review round 3, item 3: "measure it in the GEMM"): the four NN GEMMs in their in-step modes on
        # CHANNEL-MAJOR operands [C][B P] = one GEMM over all columns (B = 1, T = B P; the kernels only see a longer row pitch) against
        # the batch-major launches above.  Same buffers (bytes), T % 128 == 0 so both forms run the same tiles.
        Np = B * P
        stc_h, stc_c = torch.empty(N.helper("v100_pw_num_parts", 1, Np), hid, 2, device=dev), torch.empty(N.helper("v100_pw_num_parts", 1, Np), C, 2, device=dev)
        cm_rows = [
            ("expand fwd", L(lambda s: G(W1, s["xc"], None, None, None, None, 0, s["yh"], None, None, None, 1, st_h, B, hid, C, T, 5)),
             L(lambda s: G(W1, s["xc"], None, None, None, None, 0, s["yh"], None, None, None, 1, stc_h, 1, hid, C, Np, 5))),
            ("project fwd", L(lambda s: G(W2, s["xh"], None, ch[0], ch[1], None, 1, s["yc"], None, None, None, 1, st_c, B, C, hid, T, 5)),
             L(lambda s: G(W2, s["xh"], None, ch[0], ch[1], None, 1, s["yc"], None, None, None, 1, stc_c, 1, C, hid, Np, 5))),
            ("project bwd-data", L(lambda s: G(W2t, s["xc"], None, None, None, None, 0, s["yh"], ch[0], ch[1], s["xh2"], 4, st_h, B, hid, C, T, 13)),
             L(lambda s: G(W2t, s["xc"], None, None, None, None, 0, s["yh"], ch[0], ch[1], s["xh2"], 4, stc_h, 1, hid, C, Np, 13))),
            ("expand bwd-data", L(lambda s: G(W1t, s["xh"], s["xh2"], ch[0], ch[1], ch[2], 2, s["yc32"], None, None, s["x32"], 5, None, B, C, hid, T, 3)),
             L(lambda s: G(W1t, s["xh"], s["xh2"], ch[0], ch[1], ch[2], 2, s["yc32"], None, None, s["x32"], 5, None, 1, C, hid, Np, 3))),
        ]
        print(f"layout A/B of this library's NN GEMMs (in-step modes): batch-major [B][C][T] launch vs channel-major [C][B T] as one GEMM")
        ta = tb = 0.0
        for name, fa, fb in cm_rows:
            try:
                da, db = timeit(fa, args.iters), timeit(fb, args.iters)
                da2, db2 = timeit(fa, args.iters), timeit(fb, args.iters)          # interleaved second round
                da, db = min(da, da2), min(db, db2)
                ta += da; tb += db
                print(f"  {name:18s} batch-major {da*1e6:7.1f} us   channel-major {db*1e6:7.1f} us   ({(db/da-1)*100:+5.1f} %)")
            except Exception as ex:                                          # noqa: BLE001
                print(f"  {name:18s} ERR {type(ex).__name__}: {ex}")
        print(f"  {'sum':18s} batch-major {ta*1e6:7.1f} us   channel-major {tb*1e6:7.1f} us   ({(tb/max(ta,1e-12)-1)*100:+5.1f} %)")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
