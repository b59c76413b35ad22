"""bf16 STORAGE of the hidden tensors ("act16", include/voice100_hip.h): every *_io kernel entry point against the fp32-storage
entry point of the same arithmetic, fed the same (bf16-representable) values -- the two must agree to fp32 round-off, the only
intended difference being the rounding of a STORED output.  Shapes include a row length that is not a multiple of 8 (pitched
rows) and one that spans two 512-sample depthwise tiles."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

X, X2, Y, R = 1, 2, 4, 8            # PW_IO_*
G_, G2_, WX = 1, 2, 4               # WG_IO_*
DX, DX2, DAUX, DY = 1, 2, 4, 8      # DW_IO_*


def _native():
    from voice100_amd import _native as N
    N.load()
    return N


def pitch(T, B):
    """row pitch of a 16-bit-stored [B][C][P] tensor: the library's own rule (csrc/common.h v100_pitch16)"""
    from voice100_amd import functional as F_
    return F_.pitch16(T, B)


def to16(t):
    """fp32 [B, C, T] -> bf16 [B, C, pitch(T)] (padding filled with NaN: nothing may read it) and the rounded values as fp32."""
    B, C, T = t.shape
    out = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=t.device)
    out[:, :, :T] = t.to(torch.bfloat16)
    return out, out[:, :, :T].to(torch.float32).contiguous()


def from16(t16, T):
    return t16[:, :, :T].to(torch.float32)


SHAPES = [(2, 32, 8, 48), (3, 256, 64, 100), (2, 64, 256, 133), (1, 512, 128, 700),
          (32, 1024, 64, 512), (16, 2048, 128, 1024)]      # 512 / 1024 block tiles: persistent workgroups (2 / 4 tiles each)


@pytest.mark.parametrize("B,M,K,T", SHAPES)
def test_gemm_io_variants(cuda, B, M, K, T):
    N = _native()
    g = torch.Generator().manual_seed(M + T)
    A = (torch.randn(M, K, generator=g) / K ** 0.5).to(cuda)
    Abf = A.to(torch.bfloat16)
    x = torch.randn(B, K, T, generator=g).to(cuda)
    x16, xr = to16(x)
    x2 = torch.randn(B, K, T, generator=g).to(cuda)
    x216, x2r = to16(x2)
    xa, xb, xc = (torch.randn(K, generator=g).to(cuda) for _ in range(3))
    ea, eb = torch.rand(M, generator=g).to(cuda) + 0.5, torch.randn(M, generator=g).to(cuda)
    r = (torch.randn(B, M, T, generator=g) * 3).to(cuda)
    r16, rr = to16(r)
    parts = N.helper("v100_pw_num_parts", B, T)

    def ref(xm, ep, xin, x2in, rin):
        y = torch.empty(B, M, T, device=cuda)
        st = torch.zeros(parts, M, 2, device=cuda)
        N.call("v100_pw_gemm", A, Abf, xin, x2in, xa if xm else None, xb if xm else None, xc if xm == 2 else None, xm, y, None,
               ea if ep == 4 else None, eb if ep == 4 else None, rin, ep, st if ep in (1, 4) else None, B, M, K, T, 1)
        return y, st

    def io(xm, ep, xin, x2in, rin, mask):
        y = (torch.full((B, M, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda) if mask & Y else torch.empty(B, M, T, device=cuda))
        st = torch.zeros(parts, M, 2, device=cuda)
        N.call("v100_pw_gemm_io", Abf, xin, x2in, xa if xm else None, xb if xm else None, xc if xm == 2 else None, xm, y,
               ea if ep == 4 else None, eb if ep == 4 else None, rin, ep, st if ep in (1, 4) else None, B, M, K, T, mask)
        return y, st

    # expand forward: stats epilogue, Y stored as bf16
    y0, s0 = ref(0, 1, x, None, None)
    y1, s1 = io(0, 1, x, None, None, Y)
    assert torch.equal(from16(y1, T), y0.to(torch.bfloat16).to(torch.float32))
    assert rel_err(s1, s0) < 1e-6
    # expand forward reading the bf16 shadow of the block input (level 4): the kernel rounds X to bf16 either way -> identical
    y2, s2 = io(0, 1, x16, None, None, X | Y)
    y3, s3 = io(0, 1, xr, None, None, Y)
    assert torch.equal(from16(y2, T), from16(y3, T)) and torch.equal(s2, s3)
    # project forward: BN + ReLU6 on load of a bf16 X
    y0, s0 = ref(1, 1, xr, None, None)
    y1, s1 = io(1, 1, x16, None, None, X)
    assert rel_err(y1, y0) < 1e-6 and rel_err(s1, s0) < 1e-6
    # project backward-data: ReLU6 mask + sums from a bf16 R; fp32 and bf16 output
    y0, s0 = ref(0, 4, x, None, rr)
    y1, s1 = io(0, 4, x, None, r16, R)
    assert rel_err(y1, y0) < 1e-6 and rel_err(s1, s0) < 1e-6
    y1, s1 = io(0, 4, x, None, r16, R | Y)
    assert torch.equal(from16(y1, T), y0.to(torch.bfloat16).to(torch.float32)) and rel_err(s1, s0) < 1e-6
    # level 3: project forward with a bf16 output as well, project backward-data with a bf16 X (da3)
    y0, s0 = ref(1, 1, xr, None, None)
    y1, s1 = io(1, 1, x16, None, None, X | Y)
    assert torch.equal(from16(y1, T), y0.to(torch.bfloat16).to(torch.float32)) and rel_err(s1, s0) < 1e-6
    y0, s0 = ref(0, 4, xr, None, rr)
    y1, s1 = io(0, 4, x16, None, r16, X | R | Y)
    assert torch.equal(from16(y1, T), y0.to(torch.bfloat16).to(torch.float32)) and rel_err(s1, s0) < 1e-6
    # expand backward-data: affine of (X fp32 | bf16, X2 bf16), + residual / plain
    for ep in (5, 0):
        res = r if ep == 5 else None
        y0, _ = ref(2, ep, x, x2r, res)
        y1, _ = io(2, ep, x, x216, res, X2)
        assert rel_err(y1, y0) < 1e-6
        y0, _ = ref(2, ep, xr, x2r, res)
        y1, _ = io(2, ep, x16, x216, res, X | X2)
        assert rel_err(y1, y0) < 1e-6


@pytest.mark.parametrize("B,M,K,T", [(2, 32, 8, 48), (4, 256, 64, 100), (3, 64, 256, 133), (2, 512, 128, 704), (3, 300, 260, 133),
                                     (2, 128, 512, 192), (5, 640, 384, 77)])   # the last three: 256-row tiles, ragged rows, T tails
def test_wgrad_io_variants(cuda, B, M, K, T):
    N = _native()
    g = torch.Generator().manual_seed(M * 3 + T)
    gm = torch.randn(B, M, T, generator=g).to(cuda)
    g16, gr = to16(gm)
    g2 = torch.randn(B, M, T, generator=g).to(cuda)
    g216, g2r = to16(g2)
    xm = torch.randn(B, K, T, generator=g).to(cuda)
    x16, xr = to16(xm)
    ga, gb, gc = (torch.randn(M, generator=g).to(cuda) for _ in range(3))
    xa, xb = torch.rand(K, generator=g).to(cuda) + 0.5, torch.randn(K, generator=g).to(cuda)
    S = N.helper("v100_pw_wgrad_splits", B, M, K)

    def run(name, G, G2, gmode, Xt, xmode, mask=None):
        partial = torch.empty(S, M, K, device=cuda)
        dW = torch.empty(M, K, device=cuda)
        args = (G, G2, ga if gmode else None, gb if gmode else None, gc if gmode == 2 else None, gmode, Xt, xa if xmode else None,
                xb if xmode else None, xmode, partial, dW, S, B, M, K, T)
        if mask is None:
            N.call(name, *args, 1)
        else:
            N.call(name, *args, mask)
        return dW

    assert rel_err(run("v100_pw_wgrad_io", gm, g216, 2, xm, 0, G2_), run("v100_pw_wgrad", gm, g2r, 2, xm, 0)) < 1e-6
    assert rel_err(run("v100_pw_wgrad_io", g16, g216, 2, xm, 0, G_ | G2_), run("v100_pw_wgrad", gr, g2r, 2, xm, 0)) < 1e-6
    assert rel_err(run("v100_pw_wgrad_io", gm, None, 0, x16, 1, WX), run("v100_pw_wgrad", gm, None, 0, xr, 1)) < 1e-6
    assert rel_err(run("v100_pw_wgrad_io", g16, None, 0, x16, 1, G_ | WX), run("v100_pw_wgrad", gr, None, 0, xr, 1)) < 1e-6
    # expand gradient with X = the bf16 shadow (level 4)
    assert rel_err(run("v100_pw_wgrad_io", g16, g216, 2, x16, 0, G_ | G2_ | WX), run("v100_pw_wgrad", gr, g2r, 2, xr, 0)) < 1e-6


def test_io_gemms_repeatable(cuda):
    """The persistent expand GEMM (cross-tile prefetch), the 256-row backward-weight kernels (two register stages) and the lean
    epilogue are re-run on the same inputs: every run must reproduce the first bit for bit (a missing barrier or an early LDS
    read shows up as rare differing tiles)."""
    N = _native()
    g = torch.Generator().manual_seed(77)
    B, C, hid, T = 32, 256, 1024, 512
    P = pitch(T, B)
    x = torch.randn(B, C, T, generator=g).to(cuda)
    bf = lambda *shape: (torch.randn(*shape, generator=g) * 0.5).to(cuda).to(torch.bfloat16)
    a1, a2, dz1, da3 = bf(B, hid, P), bf(B, hid, P), bf(B, hid, P), bf(B, C, P)
    W1 = (torch.randn(hid, C, generator=g) / C ** 0.5).to(cuda).to(torch.bfloat16)
    W2t = (torch.randn(hid, C, generator=g) / hid ** 0.5).to(cuda).to(torch.bfloat16)
    ch = [torch.randn(hid, generator=g).to(cuda) for _ in range(3)]
    parts = N.helper("v100_pw_num_parts", B, T)
    S1, S2 = N.helper("v100_pw_wgrad_splits", B, hid, C), N.helper("v100_pw_wgrad_splits", B, C, hid)

    def expand_fwd():
        y = torch.empty(B, hid, P, dtype=torch.bfloat16, device=cuda)
        st = torch.empty(parts, hid, 2, device=cuda)
        N.call("v100_pw_gemm_io", W1, x, None, None, None, None, 0, y, None, None, None, 1, st, B, hid, C, T, Y)
        return y[:, :, :T].clone(), st

    def project_bwdd():
        y = torch.empty(B, hid, P, dtype=torch.bfloat16, device=cuda)
        st = torch.empty(parts, hid, 2, device=cuda)
        N.call("v100_pw_gemm_io", W2t, da3, None, None, None, None, 0, y, ch[0], ch[1], a2, 4, st, B, hid, C, T, X | R | Y)
        return y[:, :, :T].clone(), st

    def expand_wgrad():
        part, dW = torch.empty(S1, hid, C, device=cuda), torch.empty(hid, C, device=cuda)
        N.call("v100_pw_wgrad_io", dz1, a1, ch[0], ch[1], ch[2], 2, x, None, None, 0, part, dW, S1, B, hid, C, T, G_ | G2_)
        return (dW,)

    def project_wgrad():
        part, dW = torch.empty(S2, C, hid, device=cuda), torch.empty(C, hid, device=cuda)
        N.call("v100_pw_wgrad_io", da3, None, None, None, None, 0, a2, ch[0], ch[1], 1, part, dW, S2, B, C, hid, T, G_ | WX)
        return (dW,)

    for fn in (expand_fwd, project_bwdd, expand_wgrad, project_wgrad):
        first = fn()
        for _ in range(6):
            again = fn()
            for u, v in zip(first, again):
                assert torch.equal(u, v), fn.__name__


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("B,C,T,K", [(2, 8, 48, 19), (3, 6, 133, 83), (2, 4, 700, 51), (5, 12, 512, 5), (2, 4, 1100, 35), (2, 3, 260, 27)])
def test_dwconv_io_variants(cuda, B, C, T, K):
    """Depthwise kernels with 16-bit storage.  In this mode the conv's operands are bf16 tensors by definition (as under
    autocast): the transformed input relu6(bn1(a1)) -- and in backward the BN2-backward affine g' and xin -- are rounded once
    to bf16, and so are the taps (round 6: one digit, what F.conv1d sees under bf16 autocast; V100_DW_DIGITS=3 restores fp32-exact
    taps); accumulation stays fp32.  Reference: torch on exactly those rounded operands (fp32 accumulate)."""
    import torch.nn.functional as F
    N = _native()
    g = torch.Generator().manual_seed(C * 7 + T + K)
    pad = (K - 1) // 2
    a1 = (torch.randn(B, C, T, generator=g) * 2).to(cuda)
    a116, a1r = to16(a1)
    w = (torch.randn(C, K, generator=g) * 0.2).to(cuda)
    wq = _bf(w)                                                   # the taps as the kernels use them
    s1, t1 = torch.rand(C, generator=g).to(cuda) + 0.5, torch.randn(C, generator=g).to(cuda)
    G = N.helper("v100_dw_num_groups", B, C)
    pre = a1r * s1[None, :, None] + t1[None, :, None]
    xin = _bf(torch.clamp(pre, 0, 6))
    # forward: bf16 in, bf16 out, BN2 partial sums from the fp32 accumulators
    ref = F.conv1d(xin, wq[:, None, :], padding=pad, groups=C)
    y1 = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    st1 = torch.zeros(G, C, 2, device=cuda)
    N.call("v100_dwconv_fwd_train_io", a116, w, s1, t1, y1, st1, G, B, C, T, K, DX | DY)
    assert rel_err(from16(y1, T), ref) < 6e-3                     # one bf16 rounding of the stored output
    assert float((from16(y1, T) - _bf(ref)).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())    # ... and nothing more
    assert rel_err(st1.sum(0)[:, 0], ref.sum((0, 2))) < 5e-5 and rel_err(st1.sum(0)[:, 1], (ref * ref).sum((0, 2))) < 5e-5
    # fused backward: dz2 fp32 | bf16, a2 bf16, a1 bf16 -> dz1 fp32 | bf16, BN1-backward sums, dW
    dz2 = torch.randn(B, C, T, generator=g).to(cuda)
    dz216, dz2r = to16(dz2)
    a2 = torch.randn(B, C, T, generator=g).to(cuda)
    a216, a2r = to16(a2)
    ga, gb, gc = torch.rand(C, generator=g).to(cuda) + 0.5, (torch.randn(C, generator=g) * 0.3).to(cuda), (torch.randn(C, generator=g) * 0.1).to(cuda)
    for mask, dzin16, dzref in ((DX2 | DAUX, dz2, dz2), (DX | DX2 | DAUX | DY, dz216, dz2r)):
        gp = _bf(ga[None, :, None] * dzref + gb[None, :, None] * a2r + gc[None, :, None])
        xv, wv = xin.clone().requires_grad_(True), wq.clone().requires_grad_(True)
        (F.conv1d(xv, wv[:, None, :], padding=pad, groups=C) * gp).sum().backward()
        dz1r = xv.grad * ((pre > 0) & (pre < 6))
        s0r, s1r = dz1r.sum((0, 2)), (dz1r * a1r).sum((0, 2))
        dz1 = (torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda) if mask & DY else torch.empty(B, C, T, device=cuda))
        st = torch.zeros(G, C, 2, device=cuda)
        part = torch.empty(G, C, K, device=cuda)
        dw = torch.empty(C, K, device=cuda)
        N.call("v100_dwconv_bwd_io", dzin16, a216, w, ga, gb, gc, a116, s1, t1, dz1, st, part, dw, G, B, C, T, K, mask)
        if mask & DY:
            assert rel_err(from16(dz1, T), dz1r) < 6e-3
        else:
            assert rel_err(dz1, dz1r) < 5e-5
        assert rel_err(st.sum(0)[:, 0], s0r) < 1e-4 and rel_err(st.sum(0)[:, 1], s1r) < 1e-4
        assert rel_err(dw, wv.grad) < 5e-4      # fp32 summation order over B*T terms of size O(10)


@pytest.mark.parametrize("B,C,T", [(2, 8, 48), (3, 6, 133), (2, 16, 700)])
def test_chan_passes_io(cuda, B, C, T):
    """Block-boundary passes with a bf16-stored a3 / da3: y = s3*a3 + t3 (+ x), sums of (dy, dy*a3), da3 = p*dy + q*a3 + r."""
    N = _native()
    g = torch.Generator().manual_seed(C + T)
    a3 = torch.randn(B, C, T, generator=g).to(cuda)
    a316, a3r = to16(a3)
    x = torch.randn(B, C, T, generator=g).to(cuda)
    dy = torch.randn(B, C, T, generator=g).to(cuda)
    p_, q_, r_ = (torch.randn(C, generator=g).to(cuda) for _ in range(3))
    for res in (x, None):
        y = torch.empty(B, C, T, device=cuda)
        N.call("v100_chan_affine2_io", a316, res, p_, None, r_, y, B, C, T, 1)
        ref = a3r * p_[None, :, None] + r_[None, :, None] + (res if res is not None else 0)
        assert rel_err(y, ref) < 1e-6
    G = N.helper("v100_dw_num_groups", B, C)
    part = torch.zeros(G, C, 2, device=cuda)
    N.call("v100_chan_reduce2_io", dy, a316, part, G, B, C, T, 2)
    assert rel_err(part.sum(0)[:, 0], dy.sum((0, 2))) < 1e-5 and rel_err(part.sum(0)[:, 1], (dy * a3r).sum((0, 2))) < 1e-5
    da3 = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
    N.call("v100_chan_affine2_io", dy, a316, p_, q_, r_, da3, B, C, T, 6)
    ref = p_[None, :, None] * dy + q_[None, :, None] * a3r + r_[None, :, None]
    assert torch.equal(from16(da3, T), ref.to(torch.bfloat16).to(torch.float32)) or rel_err(from16(da3, T), ref) < 5e-3


def test_chan_affine2_shadow(cuda):
    N = _native()
    g = torch.Generator().manual_seed(8)
    for (B, C, T) in ((2, 12, 133), (3, 64, 512)):
        a3 = torch.randn(B, C, T, generator=g).to(cuda)
        a316, a3r = to16(a3)
        res = torch.randn(B, C, T, generator=g).to(cuda)
        p_, r_ = torch.rand(C, generator=g).to(cuda) + 0.5, torch.randn(C, generator=g).to(cuda)
        for ub, u, uref in ((1, a316, a3r), (0, a3, a3)):
            for rr in (res, None):
                y = torch.empty(B, C, T, device=cuda)
                sh = torch.full((B, C, pitch(T, B)), float("nan"), dtype=torch.bfloat16, device=cuda)
                N.call("v100_chan_affine2_shadow", u, rr, p_, r_, y, sh, B, C, T, ub)
                ref = uref * p_[None, :, None] + r_[None, :, None] + (rr if rr is not None else 0)
                assert rel_err(y, ref) < 1e-6
                assert torch.equal(from16(sh, T), y.to(torch.bfloat16).to(torch.float32))


def test_stack_with_shadow_equals_stack_without(cuda):
    """Level 4 (bf16 shadow of block outputs feeding the next block's GEMMs) against level 3 on a stack that starts with a
    stride-2 (fp32-storage) block: the GEMMs round their X operand to bf16 either way, so outputs and gradients must be IDENTICAL."""
    from voice100_amd import functional as F_
    from voice100_amd.layers import InvertedResidual
    F_.set_matmul_precision("bf16")
    keep = F_.get_activation_storage()
    try:
        outs = []
        for lv in (3, 4):
            F_.set_activation_storage(lv)
            torch.manual_seed(9)
            net = torch.nn.Sequential(InvertedResidual(16, 64, kernel_size=11, stride=2, use_residual=False),
                                      InvertedResidual(64, 64, kernel_size=19), InvertedResidual(64, 256, kernel_size=27, use_residual=False),
                                      InvertedResidual(256, 256, kernel_size=5)).to(cuda).train()
            x = torch.randn(2, 16, 313, generator=torch.Generator().manual_seed(4)).to(cuda).requires_grad_(True)
            y = net(x)
            assert (getattr(y, "_v100_shadow", None) is not None) == (lv == 4)
            (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(cuda)).sum().backward()
            outs.append((y.detach(), x.grad, [p.grad for p in net.parameters()]))
        (y0, gx0, gp0), (y1, gx1, gp1) = outs
        assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
        for a, b in zip(gp0, gp1):
            assert torch.equal(a, b)
    finally:
        F_.set_activation_storage(keep)
        F_.set_matmul_precision("fp32")


@pytest.mark.parametrize("level,cin", [(1, 64), (2, 64), (3, 64), (3, 256), (4, 256)])    # cin 256: hidden 1024 = ONE depthwise group, BatchNorm
def test_block_act16_matches_fp32_storage(cuda, level, cin):                        # finalised inside the depthwise kernels (DwFin)
    """A training-mode block at bf16 precision with the hidden tensors stored as bf16 vs the same block with fp32 storage:
    outputs, input gradient and parameter gradients within bf16 storage error of each other (and both within the bf16 bar
    of the fp32 oracle, tests/test_gpu_models.py)."""
    from voice100_amd import functional as F_
    from voice100_amd.layers import InvertedResidual
    torch.manual_seed(3)
    F_.set_matmul_precision("bf16")
    keep = F_.get_activation_storage()
    try:
        outs = []
        for lv in (0, level):
            F_.set_activation_storage(lv)
            torch.manual_seed(3)
            blk = InvertedResidual(cin, cin, kernel_size=27).to(cuda).train()
            x = torch.randn(3, cin, 203, generator=torch.Generator().manual_seed(5)).to(cuda).requires_grad_(True)
            y = blk(x)
            (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(cuda)).sum().backward()
            outs.append((y.detach(), x.grad, {k: p.grad for k, p in blk.named_parameters()},
                         torch.cat([blk.conv[1][1].running_var, blk.conv[1][1].running_mean, blk.conv[1][1].num_batches_tracked.float().reshape(1)])))
        (y0, gx0, gp0, rv0), (y1, gx1, gp1, rv1) = outs
        assert rel_err(y1, y0) < 2e-2 and rel_err(rv1, rv0) < 2e-2
        assert float((gx1 - gx0).norm() / gx0.norm()) < 5e-2
        # floor: BN1's gamma has a mathematically (almost) zero gradient here -- a per-channel rescale before the depthwise
        # conv is undone by the training-mode BN2 behind it -- so its two round-off-sized values cannot be compared relatively
        scale = max(float(v.norm()) for v in gp0.values())
        errs = {k: float((gp1[k] - gp0[k]).norm()) / max(float(gp0[k].norm()), 1e-2 * scale) for k in gp0}
        assert max(errs.values()) < 8e-2, errs
    finally:
        F_.set_activation_storage(keep)
        F_.set_matmul_precision("fp32")


@pytest.mark.parametrize("cin,k,B,T", [(320, 19, 2, 133), (256, 51, 3, 512), (64, 27, 32, 2048), (512, 83, 2, 640), (384, 35, 2, 77),
                                       (256, 19, 4, 302),
                                       # round 6: one group with the consumer-side BatchNorm finalisation (IR_FUSE_PRE) on a
                                       # time-stretched length whose pitch is whole 128-byte lines (563 -> 576), and with more partial
                                       # sums than the ahead-of-the-rows form holds (48 x 6 t-tiles = 288 > 256: the in-branch form)
                                       (256, 27, 6, 563), (256, 19, 48, 700)])
def test_block_act16_wide_shapes(cuda, cin, k, B, T):
    """Level-3 bf16 storage against fp32 storage (both with bf16 GEMM operands) on shapes that reach the kernels the small
    cases do not: 256-row backward-weight tiles with ragged rows and T tails, persistent expand GEMM (512 block tiles), one
    depthwise group with BatchNorm finalised in-kernel, time-stretched (odd) lengths."""
    _wide_case(cuda, cin, k, B, T)


def _wide_case(cuda, cin, k, B, T):
    from voice100_amd import functional as F_
    from voice100_amd.layers import InvertedResidual
    F_.set_matmul_precision("bf16")
    keep = F_.get_activation_storage()
    try:
        outs = []
        for lv in (0, 4):
            F_.set_activation_storage(lv)
            torch.manual_seed(11)
            blk = InvertedResidual(cin, cin, kernel_size=k).to(cuda).train()
            x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(5)).to(cuda).requires_grad_(True)
            y = blk(x)
            (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(cuda)).sum().backward()
            outs.append((y.detach(), x.grad, {n: p.grad for n, p in blk.named_parameters()},
                         torch.cat([blk.conv[i][1].running_var for i in (0, 1)] + [blk.conv[3].running_mean])))
            del blk, x, y
        (y0, gx0, gp0, rv0), (y1, gx1, gp1, rv1) = outs
        assert rel_err(y1, y0) < 2e-2 and rel_err(rv1, rv0) < 2e-2
        assert float((gx1 - gx0).norm() / gx0.norm()) < 5e-2
        scale = max(float(v.norm()) for v in gp0.values())
        errs = {n: float((gp1[n] - gp0[n]).norm()) / max(float(gp0[n].norm()), 1e-2 * scale) for n in gp0}
        assert max(errs.values()) < 8e-2, errs
    finally:
        F_.set_activation_storage(keep)
        F_.set_matmul_precision("fp32")


def _fuzz_seeds(default):
    import os
    env = os.environ.get("VOICE100_FUZZ_SEEDS")          # "1 2 3 ...": a wider sweep on demand
    return [int(v) for v in env.split()] if env else [default]


@pytest.mark.parametrize("seed", _fuzz_seeds(606))
def test_block_act16_random_shapes(cuda, seed):
    """The wide-shape comparison above on seeded RANDOM shapes: widths 64 ... 512, every depthwise size of the encoder (and two the
    streaming kernels do not specialise), batches up to 40, lengths 20 ... 1100 -- whatever row form, pitch, group count and
    finalisation route the shape selects."""
    import random
    rng = random.Random(seed)
    for _ in range(8):
        cin = rng.choice([64, 128, 192, 256, 320, 384, 512])
        k = rng.choice([11, 19, 27, 35, 51, 59, 67, 75, 83, 7, 33])
        B = rng.choice([2, 3, 5, 8, 16, 33, 40])
        T = rng.randint(20, 1100)
        if B * cin * 4 * T > 48 * 2048 * 600:           # keep a case under ~0.25 GB per hidden tensor
            T = max(20, (48 * 2048 * 600) // (B * cin * 4))
        try:
            _wide_case(cuda, cin, k, B, T)
        except AssertionError as e:
            raise AssertionError(f"shape (cin={cin}, k={k}, B={B}, T={T}): {e}") from e


@pytest.mark.parametrize("cin,k,B,T", [(64, 19, 2, 133), (256, 51, 3, 512), (512, 83, 2, 640), (256, 35, 2, 1023), (128, 7, 4, 77)])
def test_eval_block_bf16_storage(cuda, cin, k, B, T):
    """Inference at precision "bf16": the hidden tensors of an eval-mode block stored as bf16 (activation storage >= 1) against fp32
    storage (level 0; both with bf16 GEMM operands) and against the block in exact fp32 -- rows of <= 512, <= 768 (one wave item)
    and longer (general kernel), odd lengths."""
    from voice100_amd import functional as F_
    from voice100_amd.layers import InvertedResidual
    keep = F_.get_activation_storage()
    try:
        torch.manual_seed(cin + k)
        blk = InvertedResidual(cin, cin, kernel_size=k).to(cuda).eval()
        g = torch.Generator().manual_seed(T)
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g).to(cuda) * 0.2)
                    m.running_var.copy_((torch.rand(m.running_var.shape, generator=g) + 0.5).to(cuda))
        x = torch.randn(B, cin, T, generator=g).to(cuda)
        outs = {}
        with torch.no_grad():
            F_.set_matmul_precision("fp32")
            outs["fp32"] = blk(x)
            F_.set_matmul_precision("bf16")
            for lv in (0, 4):
                F_.set_activation_storage(lv)
                outs[lv] = blk(x)
        assert torch.isfinite(outs[4]).all()
        assert rel_err(outs[4], outs[0]) < 1.5e-2          # one extra bf16 rounding of each hidden tensor
        assert rel_err(outs[4], outs["fp32"]) < 3e-2 and rel_err(outs[0], outs["fp32"]) < 3e-2
    finally:
        F_.set_activation_storage(keep)
        F_.set_matmul_precision("fp32")
