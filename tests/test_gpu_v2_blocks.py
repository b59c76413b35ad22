"""the v2 conv blocks (dense k=5 Conv1d / ConvTranspose1d + channel LayerNorm + GELU) on the GPU
against the reference-generated golden vectors and, at the reference's real widths, against the CPU oracle."""
import pytest
import torch

from conftest import load_golden, sub, rel_err, rel_l2, assert_grads_close

pytestmark = pytest.mark.gpu

ASR_ENCODER = [[512, False, 5, 2, 2, False], [512, False, 5, 1, 2, False]]                                   # config/asr_en_base.yaml:16-18
TTS_DECODER = [[512, False, 5, 1, 2, False], [512, True, 5, 2, 2, False], [512, False, 5, 1, 2, False]]     # config/tts_en_base.yaml:20-23


@pytest.mark.parametrize("sid", ["s0", "s1", "s2"])
def test_v2_blocks_golden(cuda, sid):
    from voice100_amd.layers_v2 import get_conv_layers
    from voice100_amd import functional as F_
    F_.set_matmul_precision("fp32")
    g = load_golden("v2_blocks.npz")
    settings = [[int(v) for v in row] for row in g[sid + "/settings"]]
    m = get_conv_layers(int(g[sid + "/cin"]), settings)
    m.load_state_dict(sub(g, sid + "/state/"), strict=True)
    m = m.to(cuda)
    x = torch.from_numpy(g[sid + "/x"]).to(cuda).requires_grad_(True)
    y = m(x)
    assert y.shape == g[sid + "/y"].shape
    assert rel_err(y, g[sid + "/y"]) < 1e-4
    y.backward(torch.from_numpy(g[sid + "/gy"]).to(cuda))
    assert rel_err(x.grad, g[sid + "/gx"]) < 1e-4
    assert_grads_close({k: p.grad for k, p in m.named_parameters()}, {k: g[sid + "/grad/" + k] for k, _ in m.named_parameters()}, 1e-3)


@pytest.mark.parametrize("cin,settings,B,T", [(64, ASR_ENCODER, 2, 256), (1024, TTS_DECODER, 2, 64), (64, ASR_ENCODER, 3, 101)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_v2_blocks_real_widths_vs_oracle(cuda, cin, settings, B, T, precision):
    from oracle import cnn
    from voice100_amd.layers_v2 import get_conv_layers
    from voice100_amd import functional as F_
    torch.manual_seed(7)
    m = get_conv_layers(cin, settings)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(B, cin, T)
    params = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    xr = x.clone().requires_grad_(True)
    ref = cnn.conv_layers(xr, params, settings)
    gy = torch.randn(ref.shape)
    (ref * gy).sum().backward()
    F_.set_matmul_precision(precision)
    try:
        m = m.to(cuda)
        xg = x.to(cuda).requires_grad_(True)
        y = m(xg)
        y.backward(gy.to(cuda))
    finally:
        F_.set_matmul_precision("fp32")
    assert y.shape == ref.shape
    if precision == "fp32":
        assert rel_err(y, ref.detach()) < 1e-4
        assert rel_err(xg.grad, xr.grad) < 2e-4
        assert_grads_close({k: p.grad for k, p in m.named_parameters()}, {k: params[k].grad for k, _ in m.named_parameters()}, 1e-3)
    else:   # bf16 GEMM operands, fp32 accumulate / LayerNorm / GELU
        assert rel_l2(y, ref.detach()) < 2e-2
        assert rel_l2(xg.grad, xr.grad) < 5e-2


def test_layer_norm_gelu_kernel_edges(cuda):
    """Channel counts that are not multiples of 32, T not a multiple of 4 / 32, C up to 1024; forward + backward."""
    from voice100_amd import functional as F_
    g = torch.Generator().manual_seed(11)
    for (B, C, T) in [(1, 3, 1), (2, 5, 3), (2, 72, 33), (1, 512, 130), (2, 1000, 37), (1, 1024, 64)]:
        y = torch.randn(B, C, T, generator=g) * 2 + 0.5
        ga, be = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
        go = torch.randn(B, C, T, generator=g)
        yr, gr, br = (t.clone().requires_grad_(True) for t in (y, ga, be))
        ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(yr.transpose(1, 2), (C,), gr, br, 1e-5).transpose(1, 2))
        (ref * go).sum().backward()
        yd, gd, bd = (t.to(cuda).requires_grad_(True) for t in (y, ga, be))
        out = F_.layer_norm_gelu(yd, gd, bd, 1e-5)
        out.backward(go.to(cuda))
        assert rel_err(out, ref.detach()) < 1e-4, (B, C, T)
        assert rel_err(yd.grad, yr.grad, floor=1e-4) < 2e-4, (B, C, T)
        assert rel_err(gd.grad, gr.grad, floor=1e-3) < 2e-4 and rel_err(bd.grad, br.grad, floor=1e-3) < 2e-4, (B, C, T)


def test_conv1d_dense_shapes_and_errors(cuda):
    from voice100_amd import functional as F_
    g = torch.Generator().manual_seed(5)
    for (B, cin, cout, k, s, p, T) in [(2, 6, 10, 5, 2, 2, 17), (1, 3, 4, 3, 1, 0, 9), (2, 8, 8, 7, 3, 3, 40), (1, 4, 4, 5, 1, 2, 5)]:
        x, w, b = torch.randn(B, cin, T, generator=g), torch.randn(cout, cin, k, generator=g) * 0.3, torch.randn(cout, generator=g)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        ref = torch.nn.functional.conv1d(xr, wr, br, stride=s, padding=p)
        go = torch.randn(ref.shape, generator=g)
        (ref * go).sum().backward()
        xd, wd, bd = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
        y = F_.conv1d_dense(xd, wd, bd, stride=s, padding=p, precision="fp32")
        y.backward(go.to(cuda))
        assert y.shape == ref.shape and rel_err(y, ref.detach()) < 1e-4
        assert rel_err(xd.grad, xr.grad) < 1e-4 and rel_err(wd.grad, wr.grad) < 1e-4 and rel_err(bd.grad, br.grad) < 1e-4
    with pytest.raises(RuntimeError):
        F_.conv1d_dense(torch.randn(1, 4, 8), torch.randn(4, 4, 5))                       # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        F_.conv1d_dense(torch.randn(1, 4, 3, device=cuda), torch.randn(4, 4, 5, device=cuda))   # shorter than the kernel


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,settings,B,T", [(64, [[128, False, 5, 1, 2, True]], 3, 77), (1024, TTS_DECODER, 2, 64),
                                              (128, [[64, True, 5, 2, 2, True], [192, False, 3, 1, 1, False]], 2, 1),
                                              (64, [[64, False, 7, 1, 3, False]], 2, 300)])
def test_tap_addressed_gemm_equals_the_im2col_path(cuda, cin, settings, B, T, precision):
    """The tap-addressed GEMMs (no im2col / tap-stacked copies) contract the same operands in the same order as the
    explicit-copy path: in fp32 outputs and all gradients agree to round-off."""
    from voice100_amd.layers_v2 import get_conv_layers
    from voice100_amd import functional as F_
    torch.manual_seed(11)
    m = get_conv_layers(cin, settings).to(cuda)
    x = torch.randn(B, cin, T, device=cuda)
    F_.set_matmul_precision(precision)
    res = []
    try:
        for taps in (True, False):
            F_.USE_TAP_GEMM = taps
            m.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            y = m(xg)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(cuda)
            y.backward(gy)
            res.append((y.detach(), xg.grad, {k: p.grad.cpu() for k, p in m.named_parameters()}))
    finally:
        F_.USE_TAP_GEMM = True
        F_.set_matmul_precision("fp32")
    (y1, gx1, gp1), (y0, gx0, gp0) = res
    # bf16: a round-off-level difference in one layer's gradient flips the bf16 rounding of a few operands of the next GEMM
    ytol, gtol = (1e-6, 1e-5) if precision == "fp32" else (1e-3, 3e-3)
    assert rel_err(y1, y0) < ytol
    assert rel_err(gx1, gx0) < gtol
    assert_grads_close(gp1, gp0, gtol)
