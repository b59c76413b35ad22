"""Channel-major inference path (round 4: functional.inverted_residual_eval_cm, AudioToTextCTC._forward_eval_cm; asr.py:40-59,
62-94, 110-116 in eval mode): activations [C][B][P], one GEMM over all utterances' columns per 1x1 convolution, short rows packed
several to a depthwise wave item, hidden tensors in the GEMM operand format (bf16 / fp16).  Held to the per-module batch-major path
-- itself held to the oracle by test_gpu_models / test_gpu_kernels -- element for element, and to the fp32 path at the precision's bar."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _block(cuda, cin, cout, k, res, seed):
    from voice100_amd.layers import InvertedResidual
    torch.manual_seed(seed)
    blk = InvertedResidual(cin, cout, kernel_size=k, use_residual=res).to(cuda)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_((torch.randn(m.running_mean.shape, generator=g) * 0.2).to(cuda))
                m.running_var.copy_((torch.rand(m.running_var.shape, generator=g) + 0.5).to(cuda))
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) + 0.5).to(cuda))
                m.bias.copy_((torch.randn(m.bias.shape, generator=g) * 0.2).to(cuda))
    return blk.eval()


def _cm_block(blk, x, precision):
    from voice100_amd import functional as F_
    B, _, T = x.shape
    F_.set_matmul_precision(precision)
    try:
        yc = F_.inverted_residual_eval_cm(blk, F_.bct_to_cm(x), B, T)
        y = F_.cm_to_btc(yc, B, T).transpose(1, 2).contiguous()
    finally:
        F_.set_matmul_precision("fp32")
    return y


@pytest.mark.parametrize("k", [19, 51, 83])
def test_block_channel_major_equals_batch_major(cuda, k):
    """bf16: the same kernels on the same values in another layout -- every valid element identical, whatever the row length
    (packed short rows, one row per item, the 513 .. 768 form) and batch size (last item partly filled)."""
    from voice100_amd import functional as F_
    blk = _block(cuda, 64, 64, k, True, 10 + k)
    for B, T in ((1, 5), (3, 51), (7, 56), (33, 51), (5, 100), (2, 200), (3, 301), (2, 512), (3, 600), (2, 768)):
        g = torch.Generator().manual_seed(B * 1000 + T)
        x = torch.randn(B, 64, T, generator=g).to(cuda)
        F_.set_matmul_precision("bf16")
        try:
            with torch.no_grad():
                ref = blk(x)
        finally:
            F_.set_matmul_precision("fp32")
        got = _cm_block(blk, x, "bf16")
        assert got.shape == ref.shape
        assert torch.equal(got, ref), (k, B, T, float((got - ref).abs().max()))


@pytest.mark.parametrize("precision,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
def test_block_channel_major_vs_fp32(cuda, precision, tol):
    """Against the exact-fp32 eval path (the parity path, held to the oracle at 1e-4): bf16 at the bf16 bar; fp16 -- fp16 operands,
    fp16-stored hidden tensors, two fp16 digits per depthwise tap -- at 4e-3."""
    for (cin, cout, k, res, B, T) in ((256, 256, 35, True, 9, 51), (256, 512, 51, False, 4, 128), (512, 512, 83, True, 16, 51), (512, 512, 59, True, 3, 400)):
        blk = _block(cuda, cin, cout, k, res, k + B)
        x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(T)).to(cuda)
        with torch.no_grad():
            ref = blk(x)                                       # fp32
        got = _cm_block(blk, x, precision)
        assert rel_err(got, ref) < tol, (precision, cin, k, B, T, rel_err(got, ref))


def test_asr_eval_channel_major_model(cuda):
    """AudioToTextCTC.forward in eval mode: the channel-major path is taken for the 16-bit precisions and reproduces the per-module
    path (bf16: identical logits; fp16: at the fp16 bar, same greedy tokens where the margins allow), steps aside for hooks, for
    fp32 and for rows longer than a wave item."""
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd import functional as F_
    from voice100_amd.decode import ctc_greedy_decode
    torch.manual_seed(5)
    m = AudioToTextCTC(audio_size=64, embed_size=128, vocab_size=29, hidden_size=128).to(cuda).eval()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_((torch.randn(mod.running_mean.shape, generator=g) * 0.1).to(cuda))
                mod.running_var.copy_((torch.rand(mod.running_var.shape, generator=g) + 0.5).to(cuda))
    for B, T in ((37, 101), (2, 256), (3, 1024)):
        audio = (torch.randn(B, T, 64, generator=g) * 2 - 4).to(cuda)
        with torch.no_grad():
            ref32 = m(audio)
        outs = {}
        for prec in ("bf16", "fp16"):
            F_.set_matmul_precision(prec)
            try:
                with torch.no_grad():
                    F_.EVAL_CM = True
                    calls = []
                    orig = F_.ir_stack_eval_cm              # (the eight stride-1 blocks go through ONE stack call)
                    F_.ir_stack_eval_cm = lambda blocks, *a, **k: (calls.append(len(blocks)), orig(blocks, *a, **k))[1]
                    try:
                        outs[prec, True] = m(audio)
                    finally:
                        F_.ir_stack_eval_cm = orig
                    assert calls == [8], "the channel-major path was not taken"
                    F_.EVAL_CM = False
                    outs[prec, False] = m(audio)
            finally:
                F_.EVAL_CM = True
                F_.set_matmul_precision("fp32")
        assert outs["bf16", True].shape == ref32.shape == (B, (T + 1) // 2, 29)
        assert torch.equal(outs["bf16", True], outs["bf16", False])
        assert rel_err(outs["fp16", True], ref32) < 1e-2 and rel_err(outs["fp16", False], ref32) < 1e-2
        assert rel_err(outs["bf16", True], ref32) < 6e-2
    # fp32: never channel-major; a forward hook on an inner block: the per-module path (the hook must fire)
    audio = (torch.randn(4, 101, 64, generator=g) * 2 - 4).to(cuda)
    F_.set_matmul_precision("bf16")
    try:
        seen = []
        h = m.encoder.layers[3].register_forward_hook(lambda mod, i, o: seen.append(tuple(o.shape)))
        with torch.no_grad():
            a = m(audio)
        h.remove()
        assert seen == [(4, 64, 51)]
        with torch.no_grad():
            b = m(audio)
        assert torch.equal(a, b)
        # rows longer than 768 outputs: per-module path, still correct
        long = (torch.randn(1, 1700, 64, generator=g) * 2 - 4).to(cuda)
        with torch.no_grad():
            assert m(long).shape == (1, 850, 29)
    finally:
        F_.set_matmul_precision("fp32")


def test_asr_eval_bf16_with_autograd_on_keeps_gradients(cuda):
    """BatchNorm-frozen fine-tuning (round-4 advice): an eval-mode AudioToTextCTC with trainable parameters called with autograd ON
    must NOT take the detached channel-major path at bf16 -- the logits carry a graph and every parameter the reference's
    eval-mode forward (asr.py:110-116 under model.eval()) would reach receives a gradient; under no_grad the same call is detached
    and equals it value for value at the precision's bar."""
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    torch.manual_seed(21)
    m = AudioToTextCTC(64, 64, 29, 64).to(cuda).eval()
    x = torch.randn(3, 120, 64, device=cuda)
    keep = F_.get_matmul_precision()
    try:
        F_.set_matmul_precision("bf16")
        y = m(x)
        assert y.requires_grad and y.grad_fn is not None
        y.square().mean().backward()
        missing = [k for k, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
        assert not missing, missing
        assert any(float(p.grad.abs().max()) > 0 for p in m.encoder.parameters())
        with torch.no_grad():
            y0 = m(x)
        assert not y0.requires_grad
        assert rel_err(y0, y.detach()) < 3e-2
        for p in m.parameters():
            p.requires_grad_(False)
        assert not m(x).requires_grad                      # nothing can ask for a gradient: the channel-major path again
    finally:
        F_.set_matmul_precision(keep)


def test_graphed_forward_replays_the_eager_result(cuda):
    """infer.GraphedForward: an eval-mode forward recorded as a HIP graph gives bit-identical results on new inputs of the recorded shape,
    refuses other shapes, and works for a multi-output callable (predict)."""
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    from voice100_amd.infer import GraphedForward
    F_.set_matmul_precision("bf16")
    try:
        torch.manual_seed(0)
        m = AudioToTextCTC(64, 128, 29, 128).to(cuda).eval()
        x = torch.rand(2, 256, 64, device=cuda)
        g = GraphedForward(m, x)
        for seed in (1, 2):
            torch.manual_seed(seed)
            xi = torch.rand(2, 256, 64, device=cuda)
            with torch.no_grad():
                want = m(xi)
            assert torch.equal(g(xi), want)
        with pytest.raises(RuntimeError):
            g(torch.rand(2, 200, 64, device=cuda))
        t = AlignTextToAudioModel(vocab_size=29, hidden_size=128, use_mcep=True).to(cuda).eval()
        at = torch.randint(0, 29, (2, 40), device=cuda)
        gp = GraphedForward(t.predict, at)
        at2 = torch.randint(0, 29, (2, 40), device=cuda)
        with torch.no_grad():
            want = t.predict(at2)
        got = gp(at2)
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    finally:
        F_.set_matmul_precision("fp32")


def test_stack_call_equals_block_calls_and_follows_weight_updates(cuda):
    """ir_stack_eval_cm (one library call for a run of blocks, plan cached on the first block) == the blocks one call each, bit for
    bit; the plan is rebuilt when a parameter is written in place, replaced, or the batch / length changes."""
    from voice100_amd import functional as F_

    def same(a, b, B, T):          # the valid columns (padding columns T .. P-1 hold whatever the buffers held)
        return torch.equal(F_.cm_to_btc(a, B, T), F_.cm_to_btc(b, B, T))

    blocks = [_block(cuda, 64, 64, k, True, 70 + k) for k in (19, 27, 35)]
    F_.set_matmul_precision("bf16")
    try:
        with torch.no_grad():
            for B, T in ((3, 51), (2, 128), (3, 51)):
                g = torch.Generator().manual_seed(B * 100 + T)
                xc = F_.bct_to_cm(torch.randn(B, 64, T, generator=g).to(cuda))
                want = xc
                for blk in blocks:
                    want = F_.inverted_residual_eval_cm(blk, want, B, T)
                got = F_.ir_stack_eval_cm(blocks, xc, B, T)
                assert same(got, want, B, T)
                plan = F_._EVAL_STACK_PLANS[blocks[0]]
                assert same(F_.ir_stack_eval_cm(blocks, xc, B, T), want, B, T)
                assert F_._EVAL_STACK_PLANS[blocks[0]] is plan                           # unchanged inputs: the plan is reused
            B, T = 3, 51
            before = F_.ir_stack_eval_cm(blocks, xc, B, T).clone()
            blocks[1].conv[3].running_mean.add_(0.25)                                    # in place: the version moves
            after = F_.ir_stack_eval_cm(blocks, xc, B, T)
            want = xc
            for blk in blocks:
                want = F_.inverted_residual_eval_cm(blk, want, B, T)
            assert same(after, want, B, T) and not same(after, before, B, T)
            blocks[2].conv[2].weight = torch.nn.Parameter(blocks[2].conv[2].weight.detach() * 0.5)      # replaced: the object moves
            after2 = F_.ir_stack_eval_cm(blocks, xc, B, T)
            want = xc
            for blk in blocks:
                want = F_.inverted_residual_eval_cm(blk, want, B, T)
            assert same(after2, want, B, T) and not same(after2, after, B, T)
            # `p.data = other` (EMA / SWA swaps, checkpoint assignment) keeps the object AND its version, only the storage moves:
            # the plan bakes pointers and folded coefficients, so the key must see data_ptr() (round-5 advisor finding)
            w = blocks[0].conv[0][0].weight
            ident = (id(w), w._version)
            w.data = w.data.clone() * 2
            assert (id(w), w._version) == ident
            g1 = blocks[1].conv[0][1].weight
            g1.data = g1.data.clone() * 0.5
            after3 = F_.ir_stack_eval_cm(blocks, xc, B, T)
            want = xc
            for blk in blocks:
                want = F_.inverted_residual_eval_cm(blk, want, B, T)
            assert same(after3, want, B, T) and not same(after3, after2, B, T)
    finally:
        F_.set_matmul_precision("fp32")


def test_modules_stay_copyable_and_picklable_after_an_eval_forward(cuda):
    """The eval-stack plan holds ctypes pointer arrays (neither picklable nor deep-copyable): it must not live in the module's
    __dict__.  After an eval forward copy.deepcopy / pickle / torch.save of the blocks work and the copy computes the same."""
    import copy
    import io
    import pickle
    from voice100_amd import functional as F_
    blocks = torch.nn.ModuleList([_block(cuda, 64, 64, k, True, 90 + k) for k in (19, 27)])
    F_.set_matmul_precision("bf16")
    try:
        with torch.no_grad():
            B, T = 2, 64
            xc = F_.bct_to_cm(torch.randn(B, 64, T, generator=torch.Generator().manual_seed(5)).to(cuda))
            want = F_.cm_to_btc(F_.ir_stack_eval_cm(list(blocks), xc, B, T), B, T).clone()
            twin = copy.deepcopy(blocks)
            pickle.dumps(blocks)
            buf = io.BytesIO()
            torch.save(blocks, buf)
            assert "_v100_eval_stack_plan" not in blocks[0].__dict__
            got = F_.cm_to_btc(F_.ir_stack_eval_cm(list(twin), xc, B, T), B, T)
            assert torch.equal(got, want)
            assert twin[0] in F_._EVAL_STACK_PLANS and F_._EVAL_STACK_PLANS[twin[0]] is not F_._EVAL_STACK_PLANS[blocks[0]]
    finally:
        F_.set_matmul_precision("fp32")


def test_asr_eval_small_shapes_fuzz(cuda):
    """The shapes the latency GEMMs and the one-call stack serve (few columns: short chunks, small batches) and those just past their
    limits: channel-major forward == per-module forward, bit for bit at bf16, over a seeded spread of (B, T)."""
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd import functional as F_
    torch.manual_seed(11)
    m = AudioToTextCTC(audio_size=64, embed_size=256, vocab_size=29, hidden_size=256).to(cuda).eval()
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_((torch.randn(mod.running_mean.shape, generator=g) * 0.1).to(cuda))
                mod.running_var.copy_((torch.rand(mod.running_var.shape, generator=g) + 0.5).to(cuda))
    shapes = [(1, 16), (1, 101), (2, 256), (3, 33), (5, 77), (8, 101), (13, 64), (32, 101), (40, 50), (64, 128), (7, 511), (96, 101),
              (3, 601), (2, 1101), (5, 640)]      # rows of >= 256 output frames: pitch of whole 128-byte lines (301 -> 320, 551 -> 576)
    F_.set_matmul_precision("bf16")
    try:
        with torch.no_grad():
            for B, T in shapes:
                audio = (torch.randn(B, T, 64, generator=g) * 2 - 4).to(cuda)
                F_.EVAL_CM = True
                a = m(audio)
                F_.EVAL_CM = False
                b = m(audio)
                assert a.shape == b.shape == (B, (T + 1) // 2, 29)
                assert torch.equal(a, b), (B, T, float((a - b).abs().max()))
    finally:
        F_.EVAL_CM = True
        F_.set_matmul_precision("fp32")
