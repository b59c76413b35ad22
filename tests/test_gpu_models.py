"""Block- and model-level parity (GPU): the drop-in modules, fed the reference's golden vectors
(tests/golden/*.npz) and checked against the CPU oracle.  fp32 paths: <= 1e-4 relative on
activations / logits (max-abs error over max-abs reference), token ids bit-exact; gradients
<= 1e-3 of the model's gradient scale; bf16 GEMM operands: relaxed and stated per test."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub, rel_err, rel_l2, assert_grads_close

pytestmark = pytest.mark.gpu


def _block_ids():
    return sorted({k.split("/")[0] for k in load_golden("ir_blocks.npz")})


def _load_block(g, bid, cuda):
    from voice100_amd.layers import InvertedResidual
    cin, cout, k, s, res, B, T = (int(v) for v in g[bid + "/cfg"])
    m = InvertedResidual(cin, cout, kernel_size=k, stride=s, use_residual=bool(res))
    m.load_state_dict(sub(g, bid + "/state/"), strict=True)          # reference key names, strict
    return m.to(cuda), torch.from_numpy(g[bid + "/x"])


@pytest.mark.parametrize("bid", _block_ids())
def test_inverted_residual_golden(cuda, bid):
    g = load_golden("ir_blocks.npz")
    m, x = _load_block(g, bid, cuda)
    m.eval()
    assert rel_err(m(x.to(cuda)), g[bid + "/y_eval"]) < 1e-4
    m.train()
    xg = x.to(cuda).requires_grad_(True)
    y = m(xg)
    assert rel_err(y, g[bid + "/y_train"]) < 1e-4
    y.backward(torch.from_numpy(g[bid + "/gy"]).to(cuda))
    assert rel_err(xg.grad, g[bid + "/gx"]) < 1e-3
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert_grads_close(grads, {k: g[bid + "/grad/" + k] for k in grads}, 1e-3, what=bid + " ")
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v, g[bid + "/after/" + k]) < 1e-4, k
        if "num_batches" in k:
            assert int(v) == int(g[bid + "/after/" + k])


def test_inverted_residual_bf16_operands(cuda):
    """Throughput path: 1x1 GEMM operands rounded to bf16 (fp32 accumulate, fp32 depthwise/BN)."""
    from voice100_amd import functional as F_
    g = load_golden("ir_blocks.npz")
    F_.set_matmul_precision("bf16")
    try:
        for bid in ("b1", "b2"):
            m, x = _load_block(g, bid, cuda)
            m.eval()
            assert rel_err(m(x.to(cuda)), g[bid + "/y_eval"]) < 3e-2
            m.train()
            xg = x.to(cuda).requires_grad_(True)
            y = m(xg)
            assert rel_err(y, g[bid + "/y_train"]) < 3e-2
            y.backward(torch.from_numpy(g[bid + "/gy"]).to(cuda))
            assert rel_l2(xg.grad, g[bid + "/gx"]) < 0.12   # tiny 8-channel blocks, B*T ~ 100: worst case for bf16
    finally:
        F_.set_matmul_precision("fp32")


def _asr_from_golden(g, cuda, **kw):
    from voice100_amd.asr import AudioToTextCTC
    m = AudioToTextCTC(audio_size=64, vocab_size=29, **kw)
    m.load_state_dict(sub(g, "state/"), strict=True)
    return m.to(cuda)


def test_asr_tiny_golden(cuda):
    g = load_golden("asr_tiny.npz")
    m = _asr_from_golden(g, cuda, embed_size=32, hidden_size=32)
    audio = torch.from_numpy(g["audio"]).to(cuda)
    m.eval()
    logits = m(audio)
    assert logits.shape == g["logits_eval"].shape
    assert rel_err(logits, g["logits_eval"]) < 1e-4
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), g["argmax_eval"])              # token ids: bit-exact
    assert np.array_equal(m.output_length(torch.from_numpy(g["audio_len"])).numpy(), g["output_length"])
    m.train()
    m.decoder.layers[0].p = 0.0

    class NoAug(torch.nn.Module):
        def forward(self, a, l):
            return a, l
    m.batch_augment = NoAug()
    batch = ((audio, torch.from_numpy(g["audio_len"]).to(cuda)),
             (torch.from_numpy(g["text"]).to(cuda), torch.from_numpy(g["text_len"]).to(cuda)))
    loss = m.training_step(batch, 0)
    assert abs(float(loss) - float(g["loss_train"])) < 1e-4 * abs(float(g["loss_train"]))
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert_grads_close(grads, {k: g["grad/" + k] for k in grads}, 2e-3)
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v, g["after/" + k]) < 1e-4, k


def test_asr_c1_full_size(cuda):
    """BASELINE configs[0]: full-size asr_en_base eval forward on the reference's own CPU-runnable case."""
    from voice100_amd.asr import AudioToTextCTC
    g = load_golden("asr_c1.npz")
    torch.manual_seed(1234)
    m = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 11621661
    sums = np.array([float(v.double().sum()) for v in m.state_dict().values() if v.dtype.is_floating_point])
    # The golden logits belong to the weights torch.manual_seed(1234) produces in the build container (11.6 M parameters are not
    # shipped as a fixture).  If a torch build ever initialises differently this test must FAIL, not skip: it is the only full-size
    # configs[0] pin against the reference's own logits -- regenerate tests/golden/asr_c1.npz with tests/golden/make_golden.py then.
    assert np.allclose(sums, g["weight_sums"], rtol=0, atol=1e-9), (
        "torch's seeded initialisation differs from the one tests/golden/asr_c1.npz was generated with: the configs[0] pin cannot "
        "be checked -- regenerate the fixture with tests/golden/make_golden.py (needs /root/reference)")
    m = m.to(cuda).eval()
    logits = m(torch.from_numpy(g["audio"]).to(cuda))
    assert rel_err(logits, g["logits"]) < 1e-4
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), g["argmax"])


def test_asr_metric_shape_properties(cuda):
    """BASELINE metric size (B=32, T=1024): size-independent
    properties -- batch independence in eval mode (an utterance's logits do not depend on its batch-mates),
    shape/length arithmetic, finiteness; train-mode step produces finite grads for every parameter."""
    from voice100_amd.asr import AudioToTextCTC
    torch.manual_seed(1234)
    m = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512).to(cuda)
    audio = torch.randn(32, 1024, 64, device=cuda) * 2 - 4
    m.eval()
    full = m(audio)
    assert full.shape == (32, 512, 29) and torch.isfinite(full).all()
    part = m(audio[5:9].contiguous())
    assert rel_err(part, full[5:9]) < 1e-5
    m.train()
    text = torch.randint(1, 29, (32, 100), device=cuda)
    lens = torch.full((32,), 1024, dtype=torch.int32, device=cuda)
    loss = m.training_step(((audio, lens), (text, torch.full((32,), 100, dtype=torch.int32, device=cuda))), 0)
    loss.backward()
    assert torch.isfinite(loss)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_augment_golden(cuda):
    from voice100_amd.audio import BatchSpectrogramAugumentation, AugmentDecisions
    g = load_golden("augment.npz")
    aug = BatchSpectrogramAugumentation()
    audio = torch.from_numpy(g["audio"]).to(cuda)
    alen = torch.from_numpy(g["audio_len"])
    from oracle import augment as O

    def run(d):
        out, ln = aug(audio, alen, decisions=d)
        return out.cpu(), ln

    base_mask = O.maskaudio                                    # maskaudio always runs last (audio.py:46-49)
    for n in range(3):
        d = AugmentDecisions(); d.stretch_rate = int(g[f"timestretch{n}/rate"])
        out, ln = run(d)
        ref = torch.from_numpy(g[f"timestretch{n}/audio"]); rl = torch.from_numpy(g[f"timestretch{n}/len"])
        assert np.array_equal(ln.numpy(), rl.numpy())
        assert rel_err(out, base_mask(ref, rl)) < 1e-5
    for n in range(2):
        d = AugmentDecisions(); d.pitch_rate = float(g[f"pitchshift{n}/rate"])
        assert rel_err(run(d)[0], base_mask(torch.from_numpy(g[f"pitchshift{n}/audio"]), alen)) < 1e-5
    d = AugmentDecisions(); d.amp = float(g["ampshift/rate"])
    assert rel_err(run(d)[0], base_mask(torch.from_numpy(g["ampshift/audio"]), alen)) < 1e-5
    for n in range(3):
        d = AugmentDecisions(); d.tmask = [(int(t), int(hw), float(a)) for t, hw, a in g[f"timemask{n}/spans"]]
        assert rel_err(run(d)[0], base_mask(torch.from_numpy(g[f"timemask{n}/audio"]), alen)) < 1e-5
        t, hw, a = g[f"freqmask{n}/params"]
        d = AugmentDecisions(); d.fmask = (int(t), int(hw), float(a))
        assert rel_err(run(d)[0], base_mask(torch.from_numpy(g[f"freqmask{n}/audio"]), alen)) < 1e-5
    low, high, std = (float(v) for v in g["mixnoise/params"])
    d = AugmentDecisions(); d.noise = (low, high, std, torch.from_numpy(g["mixnoise/uniform"]))
    assert rel_err(run(d)[0], base_mask(torch.from_numpy(g["mixnoise/audio"]), alen)) < 1e-5
    d = AugmentDecisions(); d.mix = True
    assert rel_err(run(d)[0], g["mixaudio/audio"]) < 1e-5
    assert rel_err(run(AugmentDecisions())[0], g["maskaudio/audio"]) < 1e-5


def test_streaming_pipeline_config5(cuda):
    """BASELINE configs[4] shape: 1-second 16 kHz chunks -> log-mel [B,101,64] -> ConvVoiceEncoder -> logits -> greedy CTC
    tokens, all on the GPU, against the CPU oracle chain (oracle.mel -> oracle.cnn -> merge_repeated).  Each chunk is an
    independent zero-padded utterance, as the reference's forward() would treat a 101-frame input."""
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.decode import ctc_greedy_decode
    from oracle import cnn, mel as omel, intops
    g = load_golden("asr_tiny.npz")
    m = _asr_from_golden(g, cuda, embed_size=32, hidden_size=32).eval()
    state = sub(g, "state/")
    gen = torch.Generator().manual_seed(21)
    wav = torch.rand(4, 16000, generator=gen) * 2 - 1
    feats = MelSpectrogramAudioTransform().to(cuda)(wav.to(cuda))
    assert feats.shape == (4, 101, 64)
    logits = m(feats)
    assert logits.shape == (4, 51, 29)
    ids, n = ctc_greedy_decode(logits)
    ref_feats = torch.stack([torch.from_numpy(omel.log_mel(w.numpy())) for w in wav])
    ref_logits = cnn.audio_to_text_ctc_forward(ref_feats, state, training=False)
    assert rel_err(feats, ref_feats) < 1e-4 and rel_err(logits, ref_logits) < 2e-4
    for b in range(4):
        assert ids[b, :int(n[b])].cpu().tolist() == intops.merge_repeated_ids(ref_logits[b].argmax(-1).tolist())


def test_rccl_gradient_exchange_single_rank(cuda):
    """The bucketed all-reduce path (post-accumulate hooks -> flat buffer -> async RCCL all-reduce -> mean) on the
    real backend: a one-rank "nccl" (= RCCL) group on this GPU.  Gradients and the parameter update must equal the
    plain single-process step.  (The driver's multi-GPU bench is otherwise the first time this path meets a GPU.)"""
    import os
    import torch.distributed as dist
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.dist import FlatGradBuckets
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(3)
        m1 = AudioToTextCTC(64, 32, 29, 32).to(cuda)
        m2 = AudioToTextCTC(64, 32, 29, 32).to(cuda)
        m2.load_state_dict(m1.state_dict())
        for m in (m1, m2):
            m.train()
            m.batch_augment.do_timestretch = False
        g = torch.Generator().manual_seed(5)
        audio = torch.randn(4, 96, 64, generator=g).to(cuda)
        batch = ((audio, torch.full((4,), 96, dtype=torch.int32, device=cuda)),
                 (torch.randint(1, 29, (4, 10), generator=g).to(cuda), torch.full((4,), 10, dtype=torch.int32, device=cuda)))
        import random
        grads = []
        for m, force in ((m1, False), (m2, True)):
            random.seed(11); torch.manual_seed(11)
            buckets = FlatGradBuckets(m.parameters(), bucket_bytes=1 << 16, force_exchange=force)
            assert buckets.exchange == force
            buckets.begin_step()
            m.training_step(batch, 0).backward()
            buckets.finish_step()
            grads.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
            buckets.remove_hooks()
        assert len(grads[1]) == len(grads[0]) > 0
        for k in grads[0]:
            assert torch.equal(grads[0][k], grads[1][k]), k
    finally:
        if created:
            dist.destroy_process_group()


def test_two_forwards_one_backward_under_exchange(cuda):
    """Round-4 advice: with the data-parallel flat buffer registered as the stack executor's gradient arena, two forwards of the same
    model followed by ONE backward ((loss(m(x1)) + loss(m(x2))).backward()) put two stack backward nodes in one graph, both of which
    run before AccumulateGrad has set w.grad.  Each arena slice must be handed out once per step (the second node takes its own
    buffer), so that the exchanged gradient is G1 + G2 -- equal to the plain single-process gradients -- and not 2 * G2."""
    import os
    import torch.distributed as dist
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.dist import FlatGradBuckets
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(4)
        m1 = AudioToTextCTC(64, 32, 29, 32).to(cuda)
        m2 = AudioToTextCTC(64, 32, 29, 32).to(cuda)
        m2.load_state_dict(m1.state_dict())
        g = torch.Generator().manual_seed(6)
        x1 = torch.randn(4, 96, 64, generator=g).to(cuda)
        x2 = torch.randn(4, 96, 64, generator=g).to(cuda)
        grads = []
        for m, force in ((m1, False), (m2, True)):
            m.train()
            m.decoder.layers[0].p = 0.0
            buckets = FlatGradBuckets(m.parameters(), bucket_bytes=1 << 16, force_exchange=force)
            buckets.begin_step()
            (m(x1).square().mean() + m(x2).square().mean()).backward()
            buckets.finish_step()
            grads.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
            buckets.remove_hooks()
        k0 = "encoder.layers.1.conv.0.0.weight"
        scale = float(grads[0][k0].abs().max())
        for k in grads[0]:
            assert torch.allclose(grads[0][k], grads[1][k], rtol=1e-5, atol=1e-6 * max(scale, 1e-6)), k
    finally:
        if created:
            dist.destroy_process_group()


def test_config4_phone_vocab(cuda):
    """BASELINE configs[3]: asr_en_phone_base = the same network with the 71-symbol CMU vocabulary (text.py:19-31),
    32 utterances per rank.  Reduced width against the CPU oracle (forward, CTC loss, gradients), then the full-size
    per-rank step as a property check (shapes, finiteness, every parameter receives a gradient)."""
    from oracle import cnn
    from voice100_amd.asr import AudioToTextCTC
    torch.manual_seed(71)
    m = AudioToTextCTC(audio_size=64, embed_size=32, vocab_size=71, hidden_size=32)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(71)
    audio = torch.randn(3, 120, 64, generator=g) * 2 - 4
    audio_len = torch.tensor([120, 97, 64], dtype=torch.int32)
    text = torch.randint(1, 71, (3, 12), generator=g)
    text_len = torch.tensor([12, 9, 5], dtype=torch.int32)
    params = {k: v.clone().requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point and "running" not in k}
    st = dict(state); st.update(params)
    ref_logits = cnn.audio_to_text_ctc_forward(audio, st, training=True)
    ref_loss = cnn.ctc_loss_from_logits(ref_logits, audio_len, text, text_len)
    ref_loss.backward()

    m = m.to(cuda).train()
    m.decoder.layers[0].p = 0.0

    class NoAug(torch.nn.Module):
        def forward(self, a, l):
            return a, l
    m.batch_augment = NoAug()
    batch = ((audio.to(cuda), audio_len.to(cuda)), (text.to(cuda), text_len.to(cuda)))
    loss = m.training_step(batch, 0)
    loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    assert_grads_close({k: p.grad for k, p in m.named_parameters()}, {k: params[k].grad for k, _ in m.named_parameters()}, 2e-3)

    big = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=71, hidden_size=512).to(cuda).train()
    assert big.decoder.layers[1].weight.shape == (71, 512, 1)
    a = torch.randn(32, 1024, 64, device=cuda) * 2 - 4
    lens = torch.full((32,), 1024, dtype=torch.int32, device=cuda)
    tx = torch.randint(1, 71, (32, 100), device=cuda)
    loss = big.training_step(((a, lens), (tx, torch.full((32,), 100, dtype=torch.int32, device=cuda))), 0)
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in big.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    with torch.no_grad():
        big.eval()
        assert big(a[:2]).shape == (2, 512, 71)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_loop_reduces_loss(cuda, precision):
    """End-to-end training on the HIP path (augmentation off, dropout on): forward, fused CTC, backward through the block
    executors, gradient buckets, fused Adam.  Over-fitting one small batch must drive the loss down."""
    import random
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.trainer import TrainStep
    random.seed(0); torch.manual_seed(0)
    F_.set_matmul_precision(precision)
    try:
        m = AudioToTextCTC(64, 64, 29, 64, learning_rate=2e-3).to(cuda)
        m.batch_augment.do_timestretch = False
        step = TrainStep(m)
        g = torch.Generator().manual_seed(1)
        audio = (torch.randn(8, 128, 64, generator=g) * 2 - 4).to(cuda)
        batch = ((audio, torch.full((8,), 128, dtype=torch.int32, device=cuda)),
                 (torch.randint(1, 29, (8, 12), generator=g).to(cuda), torch.full((8,), 12, dtype=torch.int32, device=cuda)))
        losses = [float(step(batch)) for _ in range(60)]
    finally:
        F_.set_matmul_precision("fp32")
    assert all(l == l and l < 1e4 for l in losses)            # finite throughout
    assert sum(losses[-5:]) / 5 < 0.6 * sum(losses[:5]) / 5, (losses[:5], losses[-5:])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_weight_caches_follow_the_optimizer(cuda, precision):
    """Fused optimiser kernels update parameters without bumping tensor versions: the prepared-weight buffers (training)
    and the eval-mode cache must still see every update.  One block, a large step: outputs must change after the step,
    and an eval forward after training must match a fresh module loaded with the trained state."""
    from voice100_amd import functional as F_
    from voice100_amd.layers import InvertedResidual
    F_.set_matmul_precision(precision)
    try:
        torch.manual_seed(0)
        blk = InvertedResidual(16, 16, kernel_size=19).to(cuda)
        stack = [blk]
        opt = torch.optim.Adam(blk.parameters(), lr=0.05, fused=True)
        x = torch.randn(2, 16, 64, device=cuda)
        blk.eval()
        y_eval0 = blk(x).clone()
        blk.train()
        outs = []
        for _ in range(3):
            F_.prepare_block_weights(stack)
            y = blk(x)
            outs.append(y.detach().clone())
            opt.zero_grad()
            y.square().mean().backward()
            opt.step()
        assert not torch.allclose(outs[0], outs[1]) and not torch.allclose(outs[1], outs[2])
        blk.eval()
        y_eval1 = blk(x)
        assert not torch.allclose(y_eval0, y_eval1)
        fresh = InvertedResidual(16, 16, kernel_size=19).to(cuda).eval()
        fresh.load_state_dict(blk.state_dict())
        assert rel_err(y_eval1, fresh(x)) < 1e-6
    finally:
        F_.set_matmul_precision("fp32")


def test_fp16_inference_precision(cuda):
    """'fp16' = IEEE half MFMA operands for inference (BASELINE config 5 names fp16).  Eval forward of the ASR model and
    the TTS audio model against the fp32 path (fp16 has 3 more mantissa bits than bf16: tighter than the bf16 bar);
    training under it is refused."""
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    torch.manual_seed(5)
    asr = AudioToTextCTC(64, 64, 29, 64).to(cuda).eval()
    tts = AlignTextToAudioModel(vocab_size=29, hidden_size=64, use_mcep=False).to(cuda).eval()
    audio = torch.randn(3, 101, 64, device=cuda) * 2 - 4
    at = torch.randint(0, 29, (2, 40), device=cuda)
    with torch.no_grad():
        ref_a = asr(audio)
        ref_t = tts(at)
        errs = {}
        for prec in ("fp16", "bf16"):
            F_.set_matmul_precision(prec)
            try:
                ya = asr(audio)
                yt = tts(at)
            finally:
                F_.set_matmul_precision("fp32")
            errs[prec] = (rel_err(ya, ref_a), max(rel_err(a, b) for a, b in zip(yt, ref_t)))
    assert errs["fp16"][0] < 5e-3 and errs["fp16"][1] < 5e-3, errs
    assert errs["fp16"][0] < errs["bf16"][0] and errs["fp16"][1] < errs["bf16"][1], errs     # and better than bf16
    F_.set_matmul_precision("fp16")
    try:
        asr.train()
        with pytest.raises(RuntimeError):
            asr(audio)
    finally:
        F_.set_matmul_precision("fp32")
        asr.eval()


# bars of the bf16 / act16-level-4 step (the numeric path bench.py runs) against the fp32 oracle; see the test body
BF16_STEP_COSINE_MIN, BF16_STEP_NORM_DEV_MAX, BF16_STEP_LOSS_ERR_MAX = 0.945, 0.01, 1e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_asr_metric_shape_ragged_lengths_vs_oracle(cuda, precision):
    """BASELINE metric shape (asr_en_base, B=32 x 1024 frames) with ragged utterance lengths ~U{512..1024}:
    one fp32 training step (augmentation and dropout off) against the CPU oracle -- loss, the updated BatchNorm running
    statistics and the gradient of every parameter.  The padding frames take part in the convolutions and the batch
    statistics exactly as in the reference; only the CTC lattice sees the lengths.  bf16 = the precision bench.py runs
    (GEMM operands rounded to bf16): the same step at the relaxed bars of the bf16 path."""
    from oracle import cnn
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd import functional as F_
    torch.manual_seed(77)
    m = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(78)
    audio = torch.randn(32, 1024, 64, generator=gen) * 2 - 4
    audio_len = torch.randint(512, 1025, (32,), generator=gen, dtype=torch.int32)
    audio_len[0] = 1024
    for b in range(32):
        audio[b, int(audio_len[b]):] = float(np.log(1e-6))
    text = torch.randint(1, 29, (32, 100), generator=gen)
    text_len = torch.randint(20, 101, (32,), generator=gen, dtype=torch.int32)
    # oracle
    params = {k: v.clone().requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point and "running" not in k}
    st = dict(state); st.update(params)
    updates = cnn.BNUpdates()
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    ref_loss = cnn.audio_to_text_ctc_loss(((audio, audio_len), (text, text_len)), st, training=True, updates=updates)
    ref_grads = dict(zip(params, torch.autograd.grad(ref_loss, list(params.values()))))
    # HIP path
    def hip_step(level=None):
        mm = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512)
        mm.load_state_dict(state)
        keep = F_.get_activation_storage()
        F_.set_matmul_precision(precision)
        if level is not None:
            F_.set_activation_storage(level)
        try:
            mm = mm.to(cuda).train()
            mm.decoder.layers[0].p = 0.0
            mm.batch_augment.forward = lambda a, l: (a, l)
            ls = mm.training_step(((audio.to(cuda), audio_len.to(cuda)), (text.to(cuda), text_len.to(cuda))), 0)
            ls.backward()
        finally:
            F_.set_matmul_precision("fp32")
            F_.set_activation_storage(keep)
        return mm, ls

    if precision == "bf16":
        # The gradient of the untrained 9-block net is ill-conditioned: fp32 round-off (1e-7) already shows as 2e-3 below,
        # so bf16 operand rounding (4e-3) cannot be held to a small relative error.  What the throughput path must keep
        # is the direction: cosine similarity with the fp32 oracle gradient over all parameters.  Measured at storage level 0
        # (bf16 GEMM operands only -- what the reference's own bf16 autocast does) and at level 4 (what bench.py runs).
        recs = {}
        for level in (0, 4, 5):
            mm, ls = hip_step(level)
            got = {k: p.grad.cpu() for k, p in mm.named_parameters()}
            dot = sum(float((got[k].double() * ref_grads[k].double()).sum()) for k in ref_grads)
            n1 = sum(float(got[k].double().pow(2).sum()) for k in ref_grads) ** 0.5
            n2 = sum(float(ref_grads[k].double().pow(2).sum()) for k in ref_grads) ** 0.5
            per = {k: float((got[k].double() * ref_grads[k].double()).sum() /
                            (got[k].double().norm() * ref_grads[k].double().norm()).clamp_min(1e-30)) for k in ref_grads}
            # tensors that carry the gradient (>= 1 % of the total norm): BatchNorm shifts in front of another training-mode
            # BatchNorm have an analytically zero gradient, their "direction" is round-off on both sides
            big = [k for k in ref_grads if float(ref_grads[k].double().norm()) >= 1e-2 * n2]
            worst = min(big, key=per.get)
            recs[level] = {"loss_rel_err": abs(float(ls.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach())),
                           "grad_cosine": dot / (n1 * n2), "grad_norm_ratio": n1 / n2, "worst_big_tensor": worst,
                           "worst_big_tensor_cosine": per[worst], "n_big_tensors": len(big)}
            del mm
        # The fair bar (review round 3, item 6): what the REFERENCE's own reduced-precision run does to the same gradient.  The
        # reference trains with `--trainer.precision 16` = autocast (README.md:185-190); its CPU form is torch.autocast("cpu",
        # bfloat16) around the same oracle step: convolutions in bf16 (operands AND outputs rounded), BatchNorm / CTC as autocast
        # leaves them.  The HIP bf16 step must be no further from the fp32 gradient than that run is.
        ac_params = {k: v.clone().requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point and "running" not in k}
        ac_st = dict(state); ac_st.update(ac_params)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ac_loss = cnn.audio_to_text_ctc_loss(((audio, audio_len), (text, text_len)), ac_st, training=True)
        ac_grads = dict(zip(ac_params, torch.autograd.grad(ac_loss.float(), list(ac_params.values()))))
        n2 = sum(float(ref_grads[k].double().pow(2).sum()) for k in ref_grads) ** 0.5
        na = sum(float(ac_grads[k].double().pow(2).sum()) for k in ref_grads) ** 0.5
        dot = sum(float((ac_grads[k].double() * ref_grads[k].double()).sum()) for k in ref_grads)
        recs["reference_autocast_cpu_bf16"] = {
            "loss_rel_err": abs(float(ac_loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach())),
            "grad_cosine": dot / (na * n2), "grad_norm_ratio": na / n2}
        print("bf16 metric-shape step vs fp32 oracle, by activation-storage level:", recs)
        import json, os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(recs, open(os.path.join(out, "parity_bf16_step.json"), "w"), indent=1)
        # measured on MI355X (profiles/r03_parity_bf16_step.json: level 4 cosine 0.973, norm ratio 0.998, loss error 2e-6); asserted
        # with a 2x margin on (1 - cosine), and bf16 STORAGE (level 4) may cost at most 0.02 of cosine over bf16 operands alone (level 0)
        ac = recs["reference_autocast_cpu_bf16"]
        for level in (0, 4, 5):
            r = recs[level]
            assert r["grad_cosine"] >= ac["grad_cosine"] - 0.005, (level, r, ac)
            assert r["loss_rel_err"] <= max(2.0 * ac["loss_rel_err"], 1e-5), (level, r, ac)
            assert abs(r["grad_norm_ratio"] - 1.0) <= abs(ac["grad_norm_ratio"] - 1.0) + 0.005, (level, r, ac)
        for level in (0, 4, 5):
            r = recs[level]
            assert r["grad_cosine"] > BF16_STEP_COSINE_MIN and abs(r["grad_norm_ratio"] - 1.0) < BF16_STEP_NORM_DEV_MAX \
                and r["loss_rel_err"] < BF16_STEP_LOSS_ERR_MAX, (level, r)
        assert recs[4]["grad_cosine"] > recs[0]["grad_cosine"] - 0.02, recs
        return
    m, loss = hip_step()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-4 * abs(float(ref_loss.detach()))
    got = {k: p.grad.cpu() for k, p in m.named_parameters()}
    num = sum(float((got[k].double() - ref_grads[k].double()).pow(2).sum()) for k in ref_grads)
    den = sum(float(ref_grads[k].double().pow(2).sum()) for k in ref_grads)
    assert (num / den) ** 0.5 < 5e-3                      # all gradients, relative L2: the bar of test_gpu_fuzz (ReLU6 kinks)
    # per parameter too, but against a floor: BatchNorm biases in front of another training-mode BatchNorm have an
    # analytically zero gradient (a constant shift is normalised away) -- both sides hold round-off noise there
    floor = 1e-3 * den ** 0.5
    for k in ref_grads:
        err = float((got[k].double() - ref_grads[k].double()).norm())
        assert err < 3e-2 * max(float(ref_grads[k].double().norm()), floor), k
    sd = m.state_dict()
    for k, v in updates.items():
        if "running" in k:
            # zero-mean channels (a bias-free conv of a normalised input) hold round-off noise: floor the scale
            assert float((sd[k].cpu() - v).abs().max()) < 1e-4 * max(float(v.abs().max()), 1e-2), k
        else:
            assert int(sd[k]) == int(v), k


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_streaming_config5_full_size_vs_oracle(cuda, precision):
    """BASELINE configs[4] at full size: 32 one-second 16 kHz chunks -> log-mel -> asr_en_base encoder -> logits -> greedy
    CTC tokens, against the CPU oracle chain on the same seeded weights.  fp32: features / logits to 1e-4 and the token
    ids bit-exact for every chunk whose frame-wise argmax margins exceed the logit tolerance; fp16 (the precision the
    config names): logits at the fp16 bar, tokens equal on the chunks with margins above that bar."""
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.decode import ctc_greedy_decode
    from voice100_amd import functional as F_
    from oracle import cnn, mel as omel, intops
    torch.manual_seed(99)
    m = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512)
    gen = torch.Generator().manual_seed(100)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    wav = torch.rand(32, 16000, generator=gen) * 2 - 1
    ref_feats = torch.stack([torch.from_numpy(omel.log_mel(w.numpy())) for w in wav])
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    with torch.no_grad():
        ref_logits = cnn.audio_to_text_ctc_forward(ref_feats, state, training=False)
    F_.set_matmul_precision(precision)
    try:
        m = m.to(cuda).eval()
        with torch.no_grad():
            feats = MelSpectrogramAudioTransform().to(cuda)(wav.to(cuda))
            logits = m(feats)
            ids, n = ctc_greedy_decode(logits)
    finally:
        F_.set_matmul_precision("fp32")
    tol = 2e-4 if precision == "fp32" else 1e-2
    assert feats.shape == (32, 101, 64) and logits.shape == (32, 51, 29)
    assert rel_err(feats, ref_feats) < 1e-4 and rel_err(logits, ref_logits) < tol
    top2 = ref_logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).min(dim=1).values
    clear = margin > 4 * tol * float(ref_logits.abs().max())
    assert int(clear.sum()) >= 8
    for b in range(32):
        if clear[b]:
            assert ids[b, :int(n[b])].cpu().tolist() == intops.merge_repeated_ids(ref_logits[b].argmax(-1).tolist())


def test_augment_side_lengths_and_unit_root_gradient(cuda):
    """The augmentation pass writes the stretched lengths and the encoder's (len + 1) // 2 itself (bit-exact integers, every input
    form), and TrainStep's cached unit root gradient gives the gradients loss.backward() gives."""
    from voice100_amd import functional as F_
    from voice100_amd.audio import BatchSpectrogramAugumentation, AugmentDecisions
    aug = BatchSpectrogramAugumentation()
    g = torch.Generator().manual_seed(5)
    audio = torch.randn(7, 211, 64, generator=g).to(cuda)
    lens = torch.tensor([211, 1, 0, 210, 77, 150, 3])
    for rate in (0, 50, 99, 100, 149):
        for form in (lens, lens.to(torch.int32).to(cuda), lens.to(cuda)):
            d = AugmentDecisions(); d.stretch_rate = rate
            out, ln = aug(audio, form, decisions=d)
            want = torch.div(lens * rate, 100, rounding_mode="trunc") if rate else lens
            assert ln.dtype == form.dtype and ln.device == form.device
            assert torch.equal(ln.cpu().long(), want)
            half = F_.half_length(ln)
            if ln.is_cuda:      # the tag follows the reference's output_length contract (asr.py:81-82): the argument's device and dtype
                assert half is not None and half.dtype == ln.dtype and half.is_cuda
                assert torch.equal(half.cpu().long(), torch.div(want + 1, 2, rounding_mode="trunc"))
            else:               # host lengths carry no device tag: output_length computes on the host, like the reference
                assert half is None
            if ln.is_cuda:
                ln.add_(1)                                  # a written-to lengths tensor drops its tag
                assert F_.half_length(ln) is None
    # unit root gradient == loss.backward()
    logits = torch.randn(4, 50, 29, device=cuda, requires_grad=True)
    tgt = torch.randint(1, 29, (4, 12), device=cuda)
    il = torch.tensor([50, 40, 33, 50], dtype=torch.int32, device=cuda); tl = torch.tensor([12, 3, 7, 9], dtype=torch.int32, device=cuda)
    loss = F_.ctc_loss(logits, tgt, il, tl)
    (g0,) = torch.autograd.grad(loss, logits)
    loss = F_.ctc_loss(logits, tgt, il, tl)
    (g1,) = torch.autograd.grad(loss, logits, grad_outputs=F_.unit_grad(loss))
    loss = F_.ctc_loss(logits, tgt, il, tl)
    (g2,) = torch.autograd.grad(loss, logits, grad_outputs=torch.full((), 2.0, device=cuda))
    assert torch.equal(g0, g1) and torch.allclose(g2, 2 * g0)
    # the shortcut hands the saved buffer itself downstream: a second backward through the same node must refuse, not reuse it
    loss = F_.ctc_loss(logits, tgt, il, tl)
    torch.autograd.grad(loss, logits, grad_outputs=F_.unit_grad(loss), retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        torch.autograd.grad(loss, logits, grad_outputs=F_.unit_grad(loss))
    loss = F_.ctc_loss(logits, tgt, il, tl)                 # an explicit gradient tensor may be replayed under retain_graph
    one = torch.ones((), device=cuda)
    (g3,) = torch.autograd.grad(loss, logits, grad_outputs=one, retain_graph=True)
    (g4,) = torch.autograd.grad(loss, logits, grad_outputs=one)
    assert torch.equal(g3, g0) and torch.equal(g4, g0)


def test_augment_row_form_equals_element_form_and_writes_the_transposed_twin(cuda, monkeypatch):
    """The 64-bin row form of the augmentation pass (a wave per frame) gives the element form's values bit for bit under every
    decision, ragged frame counts included; its transposed twin is exactly transpose(1, 2), transpose_last2() hands it out while the
    batch is untouched and no gradient is wanted through it, and launches otherwise."""
    from voice100_amd import functional as F_
    from voice100_amd.audio import BatchSpectrogramAugumentation, AugmentDecisions
    import voice100_amd._native as N
    g = torch.Generator().manual_seed(11)

    def decisions(T):
        out = []
        for rate in (0, 57, 110, 149):
            for k in range(3):
                d = AugmentDecisions(); d.stretch_rate = rate
                Ts = T * rate // 100 if rate else T
                if k >= 1:
                    d.pitch_rate = 1.13; d.amp = 2.5
                    d.tmask = [(Ts // 3, 2, -7.0), (Ts - 1, 3, -6.0)]
                    d.fmask = (60, 9, -8.0)
                if k == 2:
                    d.noise = (-4.0, -1.5, 2.0, torch.rand(5, Ts, 64, generator=g)); d.mix = True
                out.append(d)
        return out

    for T in (211, 64, 1, 130):
        audio = (torch.randn(5, T, 64, generator=g) * 2 - 3).to(cuda)
        lens = torch.tensor([T, 1, 0, max(T - 1, 0), T // 2], dtype=torch.int32, device=cuda)
        for d in decisions(T):
            if d.stretch_rate and T * d.stretch_rate // 100 <= 0:
                continue
            aug = BatchSpectrogramAugumentation(); aug.emit_transposed = True
            y, ln = aug(audio, lens, decisions=d)
            yt = F_.transpose_last2(y)
            n0 = N.launch_count()
            assert F_.transpose_last2(y) is yt and N.launch_count() == n0           # the twin: no launch
            assert torch.equal(yt, y.transpose(1, 2).contiguous())
            monkeypatch.setenv("V100_AUG_ROWS", "0")
            plain = BatchSpectrogramAugumentation()
            y0, ln0 = plain(audio, lens, decisions=d)
            monkeypatch.delenv("V100_AUG_ROWS")
            assert torch.equal(y, y0) and torch.equal(ln, ln0)
            assert getattr(y0, "_v100_T", None) is None
            y.add_(1.0)                                                                # written to: the twin is stale
            yt2 = F_.transpose_last2(y)
            assert yt2 is not yt and torch.equal(yt2, y.transpose(1, 2).contiguous())
    y, _ = aug(audio, lens, decisions=AugmentDecisions())
    y.requires_grad_(True)                                                             # a gradient through the batch: the autograd op
    z = F_.transpose_last2(y)
    assert z.requires_grad and z.grad_fn is not None
