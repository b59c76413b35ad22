"""Edge cases the reference's layers accept: one-frame and shorter-than-kernel inputs, odd lengths, stride 2 on odd
lengths, even kernel sizes (generic kernels), single channel / single utterance, multi-tile rows, channel counts that
are not multiples of any tile.  InvertedResidual (train + eval + backward) against the CPU oracle, fp32 path."""
import pytest
import torch

from conftest import rel_err, assert_grads_close
from oracle import cnn

pytestmark = pytest.mark.gpu

CASES = [  # (B, cin, cout, k, stride, residual, T)
    (1, 3, 5, 3, 1, False, 1),
    (2, 4, 4, 5, 1, True, 2),
    (1, 2, 2, 83, 1, True, 7),         # far shorter than the kernel
    (3, 5, 7, 11, 2, False, 1),
    (2, 5, 7, 11, 2, False, 9),
    (2, 6, 6, 4, 1, False, 31),        # even kernel -> output length T-1, generic kernels
    (1, 1, 1, 19, 1, True, 300),
    (2, 3, 3, 27, 1, True, 1500),      # three R=8 tiles per row, ragged tail
    (2, 33, 65, 7, 1, False, 257),     # odd channel counts, partial GEMM tiles
    (5, 8, 8, 13, 1, True, 130),       # k without a specialisation
    (2, 8, 16, 6, 2, False, 50),       # even kernel, stride 2
]


@pytest.mark.parametrize("B,cin,cout,k,stride,res,T", CASES)
def test_inverted_residual_edge_shapes(cuda, B, cin, cout, k, stride, res, T):
    from voice100_amd.layers import InvertedResidual
    torch.manual_seed(1000 + T + k)
    m = InvertedResidual(cin, cout, kernel_size=k, stride=stride, use_residual=res)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.3)
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
    state = {"blk." + key: v.detach().clone() for key, v in m.state_dict().items()}
    x = torch.randn(B, cin, T)
    ref_eval = cnn.inverted_residual(x, state, "blk", k, stride, res, training=False)
    m = m.to(cuda)
    m.eval()
    y = m(x.to(cuda))
    assert y.shape == ref_eval.shape
    assert rel_err(y, ref_eval) < 1e-4
    n = B * ref_eval.shape[2]
    if n < 2:
        return                                               # batch statistics over a single value are degenerate
    params = {key: v.clone().requires_grad_(True) for key, v in state.items() if key.endswith("weight") or key.endswith("bias")}
    st = dict(state); st.update(params)
    xg = x.clone().requires_grad_(True)
    upd = cnn.BNUpdates()
    ref = cnn.inverted_residual(xg, st, "blk", k, stride, res, training=True, updates=upd)
    gy = torch.randn(ref.shape)
    ref.backward(gy)
    m.train()
    xd = x.to(cuda).requires_grad_(True)
    yt = m(xd)
    tol = 1e-4 if n >= 32 else 2e-3                          # tiny batches: rstd amplifies round-off
    assert rel_err(yt, ref) < tol
    yt.backward(gy.to(cuda))
    assert rel_err(xd.grad, xg.grad, floor=1e-4) < 10 * tol
    grads = {key: p.grad for key, p in m.named_parameters()}
    assert_grads_close(grads, {key: params["blk." + key].grad for key in grads}, 20 * tol)
    for key, v in upd.items():
        if "running" in key:
            assert rel_err(m.state_dict()[key[4:]], v) < 1e-4


def test_empty_batch_is_refused(cuda):
    from voice100_amd.layers import InvertedResidual
    m = InvertedResidual(4, 4, 5).to(cuda).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(0, 4, 16, device=cuda))
