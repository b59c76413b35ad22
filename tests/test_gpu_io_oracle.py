"""The `_io` entry points (bf16-STORED operands, include/voice100_hip.h "act16") -- the kernels the benchmark's bf16 step runs --
checked ELEMENTWISE against plain torch arithmetic in float64 on the operands as the kernels define them:

  * every 16-bit operand is the stored bf16 value; a TRANSFORMED operand (BatchNorm affine + ReLU6 on load, BatchNorm-backward
    affine of two tensors) is formed in fp32 and rounded once to bf16, because it feeds a bf16 MFMA;
  * products are exact, accumulation fp32 (reference: float64);
  * a bf16-stored OUTPUT is the fp32 result rounded once (checked to half a bf16 ulp on top of the fp32 bar).

Bar: max |got - ref| <= 2e-4 * max(1, max |ref|) per tensor (the bar of test_pw_gemm_modes for the fp32-storage kernels); an L2
norm would hide a handful of wrong elements.  tests/test_gpu_act16.py compares the same entry points with the fp32-storage HIP
kernels; this file is the independent (non-HIP) reference the round-2 review asked for.  Shapes: small ragged ones (row length
not a multiple of 8, partial tiles in every dimension) and the benchmark's own block shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

X, X2, Y, R = 1, 2, 4, 8            # PW_IO_*
G_, G2_, WX = 1, 2, 4               # WG_IO_*


def _native():
    from voice100_amd import _native as N
    N.load()
    return N


def pitch(T, B):
    """row pitch of a 16-bit-stored [B][C][P] tensor: the library's own rule (csrc/common.h v100_pitch16)"""
    from voice100_amd import functional as F_
    return F_.pitch16(T, B)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def store16(t):
    """fp32 [B, C, T] -> (bf16 [B, C, pitch(T)] with NaN padding -- nothing may read it --, the stored values as float64)"""
    B, C, T = t.shape
    out = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=t.device)
    out[:, :, :T] = t.to(torch.bfloat16)
    return out, out[:, :, :T].to(torch.float64)


def close(got, ref, what, out16=False, tol=2e-4):
    got, ref = got.double(), ref.double()
    assert torch.isfinite(got).all(), what
    bound = tol * max(1.0, float(ref.abs().max()))
    err = (got - ref).abs()
    if out16:
        err = err - ref.abs() * 2.0 ** -8          # one rounding of the stored value (half an ulp = 2^-9 relative; 2^-8 covers ties)
    worst = float(err.max())
    assert worst <= bound, f"{what}: max err {worst:.3e} > {bound:.3e} (max |ref| {float(ref.abs().max()):.3e})"


def col(v):
    return v.double()[None, :, None]


def fma32(a, b, c):
    """fmaf(a, b, c) of fp32-valued float64 tensors: the product is exact in float64, one rounding of the sum to fp32 (the kernels
    form their on-load transforms with exactly these fused operations, so the bf16 rounding that follows sees the same fp32 value)."""
    return (a * b + c).float().double()


def affine_relu6(x, a, b):           # relu6f(fmaf(x, a, b))
    return torch.clamp(fma32(x, col(a), col(b)), 0, 6)


def affine2(x, x2, a, b, c):         # fmaf(x, a, fmaf(x2, b, c))
    return fma32(x, col(a), fma32(x2, col(b), col(c)))


GEMM_SHAPES = [(2, 32, 8, 48), (3, 200, 64, 100), (2, 72, 256, 133), (1, 512, 128, 700),
               (32, 1024, 256, 512), (8, 2048, 512, 512), (8, 512, 2048, 512),      # these three: the benchmark's block shapes
               # the wave-specialised kernel (K >= 1024 with a transform on load, M % 256 == 0): odd / even k-tile counts that are
               # not multiples of its 3 or 4 register stages, a t-tile tail, a row pitch that is not T, one m-tile, K at its cap
               (2, 256, 1088, 200), (3, 512, 1024, 133), (1, 768, 1152, 64), (2, 256, 2048, 90), (1, 256, 1216, 128),
               # the overlapped-epilogue kernel (plain bf16 X, K = 256 / 512, >= 512 tiles): partial last t-tiles with T % 4 != 0 and
               # T % 8 == 0, an uneven number of tiles per workgroup (768 / 256 and 640 / 256)
               (32, 2048, 512, 379), (32, 2048, 256, 250), (20, 2048, 512, 512), (32, 1280, 256, 512),
               # rows of >= 256 samples whose pitch is a whole number of 128-byte lines and NOT the next multiple of 8 (round 6,
               # v100_pitch16: T = 300 -> 320, 563 -> 576, 520 -> 576; B > 1): every kernel family at a time-stretched length
               (3, 256, 64, 300), (4, 512, 512, 563), (5, 2048, 256, 520), (3, 256, 1024, 563), (32, 2048, 512, 563)]


@pytest.mark.parametrize("B,M,K,T", GEMM_SHAPES)
def test_pw_gemm_io_vs_float64(cuda, B, M, K, T):
    N = _native()
    g = torch.Generator().manual_seed(M * 5 + K + T)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    A = (rnd(M, K) / K ** 0.5).to(torch.bfloat16)
    Ad = A.double()
    x = rnd(B, K, T)
    x16, xs = store16(x)
    x2 = rnd(B, K, T) * 2
    x216, x2s = store16(x2)
    xa, xb, xc = torch.rand(K, generator=g).to(cuda) + 0.5, rnd(K), rnd(K) * 0.1
    ea, eb = torch.rand(M, generator=g).to(cuda) + 0.5, rnd(M)
    r = rnd(B, M, T) * 3
    r16, rs = store16(r)
    parts = N.helper("v100_pw_num_parts", B, T)

    def io(xm, ep, xin, x2in, rin, mask):
        y = (torch.full((B, M, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda) if mask & Y else
             torch.full((B, M, T), float("nan"), device=cuda))
        st = torch.zeros(parts, M, 2, device=cuda)
        N.call("v100_pw_gemm_io", A, xin, x2in, xa if xm else None, xb if xm else None, xc if xm == 2 else None, xm, y,
               ea if ep == 4 else None, eb if ep == 4 else None, rin, ep, st if ep in (1, 4) else None, B, M, K, T, mask)
        return (y[:, :, :T] if mask & Y else y), st.sum(0).double()

    def mm(xop):                                    # exact products of the bf16 operands, float64 accumulation
        return torch.einsum("mk,bkt->bmt", Ad, xop)

    def stats_close(st, y, second, what):
        scale = max(1.0, float(y.abs().sum((0, 2)).max()))
        assert float((st[:, 0] - y.sum((0, 2))).abs().max()) <= 2e-4 * scale, what + " sum"
        scale2 = max(1.0, float((y * second).abs().sum((0, 2)).max()))
        assert float((st[:, 1] - (y * second).sum((0, 2))).abs().max()) <= 2e-4 * scale2, what + " sum2"

    # expand forward (asr.py:47): plain X -- fp32 (rounded by the kernel) or the bf16 shadow --, Y stored bf16, BN1 partial sums
    ref = mm(bf(x).double())
    for xin, mask in ((x, Y), (x16, X | Y)):
        y, st = io(0, 1, xin, None, None, mask)
        close(y, ref, "expand fwd", out16=True)
        stats_close(st, ref, ref, "expand fwd stats")          # statistics come from the fp32 accumulators, not the rounded store
    # project forward (asr.py:51): BN2 + ReLU6 applied on load of the bf16 a2; Y fp32 (level <= 2) or bf16 (level >= 3)
    xt = bf(affine_relu6(xs, xa, xb).float()).double()
    ref = mm(xt)
    for mask in (X, X | Y):
        y, st = io(1, 1, x16, None, None, mask)
        close(y, ref, "project fwd", out16=bool(mask & Y))
        stats_close(st, ref, ref, "project fwd stats")
    # project backward-data: dz2 = (W3^T da3) * [0 < bn2(a2) < 6], sums (dz2, dz2 * a2); X = da3 fp32 | bf16, R = a2 bf16
    pre = fma32(rs, col(ea), col(eb))                         # the epilogue's own fmaf(R, ea, eb): the mask is then exact
    keep = ((pre > 0) & (pre < 6)).double()
    edge = torch.zeros_like(keep, dtype=torch.bool)
    for xin, xop, mask in ((x, bf(x).double(), R), (x, bf(x).double(), R | Y), (x16, xs, X | R | Y)):
        y, st = io(0, 4, xin, None, r16, mask)
        ref = mm(xop) * keep
        yy = torch.where(edge, ref.to(y.dtype), y)
        close(yy, ref, "project bwd-data", out16=bool(mask & Y))
        if not bool(edge.any()):
            stats_close(st, ref, rs, "project bwd-data stats")
    # expand backward-data (+ residual gradient): X' = p*dz1 + q*a1 + r rounded to bf16, dx = W1^T X' (+ dy), fp32 out
    for xin, xop, mask in ((x, x.double(), X2), (x16, xs, X | X2)):
        xt = bf(affine2(xop, x2s, xa, xb, xc).float()).double()
        for ep, res in ((5, r), (0, None)):
            y, _ = io(2, ep, xin, x216, res, mask)
            ref = mm(xt) + (res.double() if res is not None else 0)
            close(y, ref, "expand bwd-data")


@pytest.mark.parametrize("B,M,K,T", [(2, 32, 8, 48), (4, 200, 64, 100), (3, 64, 256, 133), (3, 300, 260, 133), (5, 640, 384, 77),
                                     (3, 256, 64, 300), (4, 512, 2048, 563), (4, 2048, 512, 563),      # pitch of whole 128-byte lines (see GEMM_SHAPES)
                                     (32, 1024, 256, 512), (16, 2048, 512, 512), (16, 512, 2048, 512),
                                     # the wave-specialised project gradient (M % 256 == 0, K % 128 == 0): T % 64 != 0 with a
                                     # pitch that is not T, a single step per split, odd step counts, more splits than steps allow
                                     (3, 256, 384, 133), (5, 256, 128, 77), (2, 512, 256, 64), (7, 256, 1024, 192), (1, 256, 128, 40)])
def test_pw_wgrad_io_vs_float64(cuda, B, M, K, T):
    N = _native()
    g = torch.Generator().manual_seed(M * 3 + T + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    gm = rnd(B, M, T)
    g16, gs = store16(gm)
    g2 = rnd(B, M, T)
    g216, g2s = store16(g2)
    xm_ = rnd(B, K, T)
    x16, xs = store16(xm_)
    ga, gb, gc = rnd(M), rnd(M), rnd(M) * 0.1
    xa, xb = torch.rand(K, generator=g).to(cuda) + 0.5, rnd(K)
    S = N.helper("v100_pw_wgrad_splits", B, M, K)

    def run(G, G2, gmode, Xt, xmode, mask):
        partial = torch.empty(S, M, K, device=cuda)
        dW = torch.full((M, K), float("nan"), device=cuda)
        N.call("v100_pw_wgrad_io", G, G2, ga if gmode else None, gb if gmode else None, gc if gmode == 2 else None, gmode, Xt,
               xa if xmode else None, xb if xmode else None, xmode, partial, dW, S, B, M, K, T, mask)
        return dW

    def ref(gop, xop):
        return torch.einsum("bmt,bkt->mk", gop, xop)

    # expand weight gradient (asr.py:47 backward): G' = p*dz1 + q*a1 + r rounded to bf16; X = block input fp32 | bf16 shadow
    for Gin, gop, Xin, xop, mask in ((gm, gm.double(), xm_, bf(xm_).double(), G2_), (g16, gs, xm_, bf(xm_).double(), G_ | G2_),
                                     (g16, gs, x16, xs, G_ | G2_ | WX)):
        gt = bf(affine2(gop, g2s, ga, gb, gc).float()).double()
        # B*T products per element: the bar scales with the accumulated magnitude like the GEMM's does with max |ref|
        close(run(Gin, g216, 2, Xin, 0, mask), ref(gt, xop), "expand wgrad", tol=3e-4)
    # project weight gradient (asr.py:51 backward): G = da3 fp32 | bf16, X' = relu6(bn2(a2)) rounded to bf16
    xt = bf(affine_relu6(xs, xa, xb).float()).double()
    for Gin, gop, mask in ((gm, bf(gm).double(), WX), (g16, gs, G_ | WX)):
        close(run(Gin, None, 0, x16, 1, mask), ref(gop, xt), "project wgrad", tol=3e-4)


@pytest.mark.parametrize("B,C,T", [(2, 8, 48), (3, 6, 133), (32, 512, 512), (5, 12, 300), (4, 64, 563)])
def test_chan_passes_io_vs_float64(cuda, B, C, T):
    """Block-boundary passes on a bf16-stored a3 / da3 (asr.py:52-59 and their backward), elementwise."""
    N = _native()
    g = torch.Generator().manual_seed(C + T)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    a3 = rnd(B, C, T) * 2
    a316, a3s = store16(a3)
    x, dy = rnd(B, C, T), rnd(B, C, T)
    p_, q_, r_ = rnd(C), rnd(C), rnd(C)
    for res in (x, None):
        ref = a3s * col(p_) + col(r_) + (res.double() if res is not None else 0)
        y = torch.full((B, C, T), float("nan"), device=cuda)
        N.call("v100_chan_affine2_io", a316, res, p_, None, r_, y, B, C, T, 1)
        close(y, ref, "block output", tol=2e-6)
        y = torch.full((B, C, T), float("nan"), device=cuda)
        sh = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
        N.call("v100_chan_affine2_shadow", a316, res, p_, r_, y, sh, B, C, T, 1)
        close(y, ref, "block output (shadow form)", tol=2e-6)
        assert torch.equal(sh[:, :, :T], y.to(torch.bfloat16))                    # the shadow is exactly the rounded output
    G = N.helper("v100_dw_num_groups", B, C)
    part = torch.zeros(G, C, 2, device=cuda)
    N.call("v100_chan_reduce2_io", dy, a316, part, G, B, C, T, 2)
    s = part.sum(0).double()
    n = B * T
    assert float((s[:, 0] - dy.double().sum((0, 2))).abs().max()) <= 2e-6 * n
    assert float((s[:, 1] - (dy.double() * a3s).sum((0, 2))).abs().max()) <= 2e-6 * n * 4
    da3 = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    N.call("v100_chan_affine2_io", dy, a316, p_, q_, r_, da3, B, C, T, 6)
    close(da3[:, :, :T], col(p_) * dy.double() + col(q_) * a3s + col(r_), "BatchNorm-3 backward", out16=True, tol=2e-6)


@pytest.mark.parametrize("B,C,T,K", [(2, 8, 48, 19), (3, 6, 133, 83), (2, 4, 700, 51), (2, 4, 1100, 35), (32, 64, 512, 59), (32, 64, 512, 27),
                                     (3, 8, 763, 83), (5, 4, 768, 19), (2, 4, 520, 59), (6, 4, 513, 67), (9, 3, 379, 75),
                                     (1, 1, 5, 7), (2, 3, 8, 5), (4, 2, 1, 5), (7, 5, 257, 11), (33, 2, 40, 17)])      # tiny / degenerate rows, B > 32
def test_dwconv_io_vs_float64(cuda, B, C, T, K):
    """Depthwise forward / fused backward with 16-bit storage (asr.py:49 and its autograd), elementwise against float64 on the
    operands as the kernels define them (transformed data operand and taps rounded once to bf16 -- F.conv1d under bf16 autocast;
    V100_DW_DIGITS=3 restores fp32-exact taps)."""
    import torch.nn.functional as F
    N = _native()
    DX, DX2, DAUX, DY = 1, 2, 4, 8
    g = torch.Generator().manual_seed(C * 7 + T + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    pad = (K - 1) // 2
    a116, a1s = store16(rnd(B, C, T) * 2)
    w = rnd(C, K) * 0.2
    s1, t1 = torch.rand(C, generator=g).to(cuda) + 0.5, rnd(C)
    G = N.helper("v100_dw_num_groups", B, C)
    pre = fma32(a1s, col(s1), col(t1))
    xin = bf(torch.clamp(pre, 0, 6).float()).double()
    wq = bf(w).double()                                       # the taps as the kernels use them
    ref = F.conv1d(xin, wq[:, None, :], padding=pad, groups=C)
    y = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    st = torch.zeros(G, C, 2, device=cuda)
    N.call("v100_dwconv_fwd_train_io", a116, w, s1, t1, y, st, G, B, C, T, K, DX | DY)
    close(y[:, :, :T], ref, "depthwise fwd", out16=True)
    s = st.sum(0).double()
    assert float((s[:, 0] - ref.sum((0, 2))).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().sum((0, 2)).max()))
    assert float((s[:, 1] - (ref * ref).sum((0, 2))).abs().max()) <= 2e-4 * max(1.0, float((ref * ref).sum((0, 2)).max()))
    dz216, dz2s = store16(rnd(B, C, T))
    a216, a2s = store16(rnd(B, C, T))
    ga, gb, gc = torch.rand(C, generator=g).to(cuda) + 0.5, rnd(C) * 0.3, rnd(C) * 0.1
    gp = bf(affine2(dz2s, a2s, ga, gb, gc).float()).double()
    xv, wv = xin.clone().requires_grad_(True), bf(w).double().clone().requires_grad_(True)
    (F.conv1d(xv, wv[:, None, :], padding=pad, groups=C) * gp).sum().backward()
    mask = ((pre > 0) & (pre < 6)).double()
    edge = torch.zeros_like(mask, dtype=torch.bool)          # pre is the kernel's own fmaf: no kink ambiguity
    dz1r = xv.grad * mask
    dz1 = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    st = torch.zeros(G, C, 2, device=cuda)
    part = torch.empty(G, C, K, device=cuda)
    dw = torch.empty(C, K, device=cuda)
    N.call("v100_dwconv_bwd_io", dz216, a216, w, ga, gb, gc, a116, s1, t1, dz1, st, part, dw, G, B, C, T, K, DX | DX2 | DAUX | DY)
    got = torch.where(edge, dz1r.to(torch.bfloat16), dz1[:, :, :T])
    close(got, dz1r, "depthwise bwd-data", out16=True)
    close(dw, wv.grad, "depthwise wgrad", tol=3e-4)
    if not bool(edge.any()):
        s = st.sum(0).double()
        assert float((s[:, 0] - dz1r.sum((0, 2))).abs().max()) <= 2e-4 * max(1.0, float(dz1r.abs().sum((0, 2)).max()))
        assert float((s[:, 1] - (dz1r * a1s).sum((0, 2))).abs().max()) <= 2e-4 * max(1.0, float((dz1r * a1s).abs().sum((0, 2)).max()))


@pytest.mark.parametrize("B,C,T,K", [(2, 8, 48, 19), (3, 6, 133, 83), (32, 64, 512, 59), (32, 16, 512, 83), (9, 3, 379, 75), (7, 5, 257, 11),
                                     (32, 8, 500, 35), (1, 2, 5, 7), (29, 4, 64, 67)])
def test_dwconv_bwd_da1_vs_float64(cuda, B, C, T, K):
    """The fused depthwise backward in its FINISHED-GRADIENT form (v100_dwconv_bwd_da1_io: what the 16-bit training step runs on every
    stride-1 block, autograd of asr.py:45-49): the kernel finalises BatchNorm 1's backward from its own sums and writes
    da1 = p dz1 + q a1 + r.  Elementwise against float64 on the operands as the kernels define them: the sums, (p, q, r), dgamma / dbeta,
    the weight gradient, and da1 formed from the kernel's OWN (p, q, r) and the bf16-rounded dz1 (the value its consumers used to
    re-derive on load), rounded once."""
    import torch.nn.functional as F
    N = _native()
    g = torch.Generator().manual_seed(C * 11 + T + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    pad = (K - 1) // 2
    a116, a1s = store16(rnd(B, C, T) * 2)
    w = rnd(C, K) * 0.2
    s1, t1 = torch.rand(C, generator=g).to(cuda) + 0.5, rnd(C)
    pre = fma32(a1s, col(s1), col(t1))
    xin = bf(torch.clamp(pre, 0, 6).float()).double()
    dz216, dz2s = store16(rnd(B, C, T))
    a216, a2s = store16(rnd(B, C, T))
    ga, gb, gc = torch.rand(C, generator=g).to(cuda) + 0.5, rnd(C) * 0.3, rnd(C) * 0.1
    gp = bf(affine2(dz2s, a2s, ga, gb, gc).float()).double()
    xv, wv = xin.clone().requires_grad_(True), bf(w).double().clone().requires_grad_(True)
    (F.conv1d(xv, wv[:, None, :], padding=pad, groups=C) * gp).sum().backward()
    mask = ((pre > 0) & (pre < 6)).double()
    dz1r = xv.grad * mask
    gamma, mean, rstd = torch.rand(C, generator=g).to(cuda) + 0.5, rnd(C) * 0.3, torch.rand(C, generator=g).to(cuda) + 0.5
    da1 = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    st = torch.zeros(1, C, 2, device=cuda)
    dw = torch.empty(C, K, device=cuda)
    pqr = torch.empty(3, C, device=cuda)
    dga, dbe = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    N.call("v100_dwconv_bwd_da1_io", dz216, a216, w, ga, gb, gc, a116, s1, t1, da1, st, dw, gamma, mean, rstd, pqr, dga, dbe, B, C, T, K)
    close(dw, wv.grad, "depthwise wgrad (da1 form)", tol=3e-4)
    s = st[0].double()
    S0, S1 = dz1r.sum((0, 2)), (dz1r * a1s).sum((0, 2))
    assert float((s[:, 0] - S0).abs().max()) <= 2e-4 * max(1.0, float(dz1r.abs().sum((0, 2)).max()))
    assert float((s[:, 1] - S1).abs().max()) <= 2e-4 * max(1.0, float((dz1r * a1s).abs().sum((0, 2)).max()))
    # BatchNorm backward from the kernel's own sums (float64 like the kernel: voice100_amd/csrc/depthwise_common.h dw_finalize_bwd_pre)
    n = float(B * T)
    mu, rs, gm = mean.double(), rstd.double(), gamma.double()
    dg = rs * (s[:, 1] - mu * s[:, 0])
    pp = gm * rs
    qq = -gm * rs * rs * dg / n
    rr = -pp * s[:, 0] / n - qq * mu
    for got, ref, what in ((pqr[0], pp, "p"), (pqr[1], qq, "q"), (pqr[2], rr, "r"), (dga, dg, "dgamma"), (dbe, s[:, 0], "dbeta")):
        assert float((got.double() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max())), what
    # da1 = fmaf(dz1, p, fmaf(a1, q, r)) on the STORED (bf16) dz1 and the kernel's fp32 coefficients, rounded once to bf16
    dz1_16 = bf(dz1r.float()).double()
    ref_da1 = fma32(dz1_16, col(pqr[0]), fma32(a1s, col(pqr[1]), col(pqr[2])))
    got = da1[:, :, :T].double()
    assert torch.isfinite(got).all()
    # a dz1 that sits on a bf16 rounding tie (fp32 accumulation order) may round the other way: one bf16 ulp of dz1, times |p|
    slack = (dz1r.abs() * 2.0 ** -7) * col(pqr[0]).abs() + ref_da1.abs() * 2.0 ** -8 + 2e-4 * max(1.0, float(ref_da1.abs().max()))
    bad = ((got - ref_da1).abs() > slack)
    assert not bool(bad.any()), f"da1: {int(bad.sum())} elements off, worst {float((got - ref_da1).abs().max()):.3e}"


# (M, K, N): one-matrix GEMMs of the channel-major inference path -- small enough in 256 x 128 tiles for the LATENCY form
# (pw_gemm_lat_kernel: 64 x 64 tiles, four k-tiles of loads in flight): one k-tile, k-tile counts that are and are not multiples of its
# register stages, a last column tile that is partly / wholly past N, the models' own widths
EVAL_CM_SHAPES = [(64, 64, 8), (128, 320, 200), (256, 64, 56), (192, 448, 136), (512, 2048, 256), (2048, 512, 264), (256, 1024, 1792)]


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("M,K,N", EVAL_CM_SHAPES)
def test_eval_gemms_one_matrix_vs_float64_and_vs_batched(cuda, M, K, N, fmt):
    """The two eval-mode GEMMs (epilogue 2: relu6(ea acc + eb) stored 16-bit from an fp32 X; epilogue 3: ea acc + eb (+ R) in fp32 from a
    16-bit X) called as ONE matrix (B = 1, T = N: what v100_ir_fwd_eval issues on channel-major activations): elementwise against float64
    on the rounded operands, and BIT FOR BIT against the same columns issued as two utterances (B = 2, T = N / 2 -- that call takes the
    throughput kernels whatever the size, so the latency form must accumulate and round exactly as they do)."""
    N_ = _native()
    f16 = fmt == "fp16"
    dt = torch.float16 if f16 else torch.bfloat16
    F16 = 16
    g = torch.Generator().manual_seed(M + 3 * K + 7 * N)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    A = (rnd(M, K) / K ** 0.5).to(dt)
    Ad = A.double()
    x = rnd(1, K, N)
    ea, eb = torch.rand(M, generator=g).to(cuda) + 0.5, rnd(M)
    r = rnd(1, M, N) * 2
    half = N // 2 if (N // 2) % 8 == 0 else None          # the two-utterance split needs a pitch equal to the row length

    def split(t):            # [1, C, N] -> [2, C, N / 2]: the same columns as two utterances
        return torch.stack((t[0, :, :half], t[0, :, half:]), 0).contiguous()

    def join(t):
        return torch.cat((t[0], t[1]), -1)[None]

    # expand: X fp32 -> h1 16-bit
    def expand(xin, B, T):
        y = torch.full((B, M, T), float("nan"), dtype=dt, device=cuda)
        N_.call("v100_pw_gemm_io", A, xin, None, None, None, None, 0, y, ea, eb, None, 2, None, B, M, K, T, Y | (F16 if f16 else 0))
        return y
    y1 = expand(x, 1, N)
    ref = torch.clamp(fma32(torch.einsum("mk,bkt->bmt", Ad, x.to(dt).double()).float().double(), col(ea), col(eb)), 0, 6)
    got, refd = y1.double(), ref
    assert torch.isfinite(got).all()
    ulp = 2.0 ** (-10 if f16 else -8)
    err = (got - refd).abs() - refd.abs() * ulp
    assert float(err.max()) <= 2e-4 * max(1.0, float(refd.abs().max())), ("expand", float(err.max()))
    if half:
        assert torch.equal(y1, join(expand(split(x), 2, half))), "expand: latency form != throughput kernels"
    # project: X 16-bit -> y fp32 (+ residual)
    x16 = x.to(dt).contiguous()

    def project(xin, rin, B, T):
        y = torch.full((B, M, T), float("nan"), device=cuda)
        N_.call("v100_pw_gemm_io", A, xin, None, None, None, None, 0, y, ea, eb, rin, 3, None, B, M, K, T, X | (F16 if f16 else 0))
        return y
    for res in (None, r):
        y3 = project(x16, res, 1, N)
        ref = fma32(torch.einsum("mk,bkt->bmt", Ad, x16.double()).float().double(), col(ea), col(eb)) + (res.double() if res is not None else 0)
        close(y3, ref, "project eval")
        if half:
            assert torch.equal(y3, join(project(split(x16), split(res) if res is not None else None, 2, half))), "project: latency form != throughput kernels"
