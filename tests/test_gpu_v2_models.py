"""The reference's v2 models (voice100/models/_asr_v2.py AudioToAlignText, _tts_v2.py AlignTextToAudio) built on the HIP
conv blocks, against the reference-generated golden vectors (tests/golden/v2_models.npz) and the CPU oracle."""
import pytest
import torch

from conftest import load_golden, sub, rel_err, rel_l2, assert_grads_close

pytestmark = pytest.mark.gpu


def _asr(g, layers, cuda):
    from voice100_amd.models_v2 import AudioToAlignText
    pre = f"asr{layers}/"
    settings = [[int(v) for v in row] for row in g[pre + "settings"]]
    audio_size, nl, hidden, vocab = (int(v) for v in g[pre + "hp"])
    m = AudioToAlignText(audio_size=audio_size, encoder_settings=settings, decoder_num_layers=nl, decoder_hidden_size=hidden, vocab_size=vocab)
    m.load_state_dict(sub(g, pre + "state/"), strict=True)
    return m.to(cuda), pre


@pytest.mark.parametrize("layers", [2, 1])
def test_v2_asr_golden(cuda, layers):
    from voice100_amd import functional as F_
    F_.set_matmul_precision("fp32")
    g = load_golden("v2_models.npz")
    m, pre = _asr(g, layers, cuda)
    t = lambda k: torch.from_numpy(g[pre + k]).to(cuda)
    m.eval()
    with torch.no_grad():
        logits, lens = m(t("audio"), t("audio_len"))
        loss = m._calc_batch_loss(((t("audio"), t("audio_len")), (t("text"), t("text_len"))))
    assert torch.equal(lens.cpu(), torch.from_numpy(g[pre + "logits_len"]))
    assert rel_err(logits, g[pre + "logits"]) < 1e-4
    assert abs(float(loss) - float(g[pre + "eval_loss"])) < 1e-4 * float(g[pre + "eval_loss"])
    if layers == 1:
        m.train()
        m.batch_augment = torch.nn.Identity()
        m.batch_augment.forward = lambda a, l: (a, l)
        loss = m.training_step(((t("audio"), t("audio_len")), (t("text"), t("text_len"))))
        loss.backward()
        assert abs(float(loss.detach()) - float(g[pre + "train_loss"])) < 1e-4 * float(g[pre + "train_loss"])
        assert_grads_close({k: p.grad for k, p in m.named_parameters()}, {k: g[pre + "grad/" + k] for k, _ in m.named_parameters()}, 1e-3)


def test_v2_asr_best_path_matches_oracle(cuda):
    """ctc_best_path of the v2 model: greedy ids without text; with text the (hist, path) of the reference's per-utterance
    numpy loop (oracle/intops), computed on the model's own log-probabilities."""
    from oracle import intops
    g = load_golden("v2_models.npz")
    m, pre = _asr(g, 2, cuda)
    m.eval()
    t = lambda k: torch.from_numpy(g[pre + k]).to(cuda)
    ids = m.ctc_best_path(t("audio"), t("audio_len"))
    with torch.no_grad():
        logits, lens = m(t("audio"), t("audio_len"))
        lp = torch.log_softmax(logits, dim=-1)
    assert torch.equal(ids, lp.argmax(-1))
    score, hist, path, out_len = m.ctc_best_path(audio_len=lens, text=t("text"), text_len=t("text_len"), logits=lp)
    assert torch.equal(out_len, lens.cpu())
    lp_c, text, text_len = lp.cpu().numpy(), g[pre + "text"], g[pre + "text_len"]
    for i in range(lp_c.shape[1]):
        n = int(lens[i])
        s, h, p = intops.ctc_best_path(lp_c[:n, i], text[i, :min(n, int(text_len[i]))])
        assert hist[i, :n].cpu().tolist() == h.tolist()
        assert path[i, :n].cpu().tolist() == p.tolist()
        assert not hist[i, n:].any() and not path[i, n:].any()
        assert abs(float(score[i]) - float(s)) < 1e-3 * max(1.0, abs(float(s)))


def _tts(g, layers, cuda):
    from voice100_amd.models_v2 import AlignTextToAudio
    pre = f"tts{layers}/"
    settings = [[int(v) for v in row] for row in g[pre + "settings"]]
    vocab, S, CA, nl, hidden = (int(v) for v in g[pre + "hp"])
    m = AlignTextToAudio(vocab_size=vocab, logspc_size=S, codeap_size=CA, encoder_num_layers=nl, encoder_hidden_size=hidden, decoder_settings=settings)
    m.load_state_dict(sub(g, pre + "state/"), strict=True)
    return m.to(cuda), pre


@pytest.mark.parametrize("layers", [2, 1])
def test_v2_tts_golden(cuda, layers):
    from voice100_amd import functional as F_
    F_.set_matmul_precision("fp32")
    g = load_golden("v2_models.npz")
    m, pre = _tts(g, layers, cuda)
    t = lambda k: torch.from_numpy(g[pre + k]).to(cuda)
    batch = ((t("f0"), t("f0_len"), t("logspc"), t("codeap")), (t("aligntext"), t("aligntext_len")))
    m.eval()
    with torch.no_grad():
        fw = m(t("aligntext"), t("aligntext_len"))
        pr = m.predict(t("aligntext"), t("aligntext_len"))
        ls = m._calc_batch_loss(batch)
    for n, v in enumerate(fw):
        assert v.shape == g[pre + f"fwd{n}"].shape
        assert rel_err(v, g[pre + f"fwd{n}"]) < 1e-4
    # the gates are sign tests on the logits: compare where the golden logit is not within rounding of zero
    for n, (v, gate) in enumerate(zip(pr, (g[pre + "fwd0"], None, g[pre + "fwd3"]))):
        ref = torch.from_numpy(g[pre + f"pred{n}"])
        if gate is None:
            assert rel_err(v, ref) < 1e-4
        else:
            sure = torch.from_numpy(abs(gate) > 1e-4)
            assert rel_err(v.cpu() * sure, ref * sure) < 1e-4
    for a, b in zip(ls, g[pre + "eval_losses"]):
        assert abs(float(a) - float(b)) < 1e-4 * abs(float(b))
    if layers == 1:
        m.train()
        loss = m.training_step(batch)
        loss.backward()
        assert abs(float(loss.detach()) - float(g[pre + "train_loss"])) < 1e-4 * float(g[pre + "train_loss"])
        want = {k[len(pre + "grad/"):]: v for k, v in g.items() if k.startswith(pre + "grad/")}
        assert_grads_close({k: p.grad for k, p in m.named_parameters() if k in want}, want, 1e-3)
        assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("norm."))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_v2_models_base_config_train_step(cuda, precision):
    """config/asr_en_base.yaml / tts_en_base.yaml widths: one optimizer step each, loss finite and decreasing on a fixed
    batch; the fp32 forward agrees with the CPU oracle on the same weights."""
    import random
    from oracle import cnn
    from voice100_amd import functional as F_
    from voice100_amd.models_v2 import AudioToAlignText, AlignTextToAudio
    enc = [[512, False, 5, 2, 2, False], [512, False, 5, 1, 2, False]]
    dec = [[512, False, 5, 1, 2, False], [512, True, 5, 2, 2, False], [512, False, 5, 1, 2, False]]
    torch.manual_seed(3)
    random.seed(3)
    F_.set_matmul_precision(precision)
    try:
        asr = AudioToAlignText(audio_size=64, encoder_settings=enc, decoder_num_layers=2, decoder_hidden_size=512, vocab_size=29)
        state = {k: v.detach().clone() for k, v in asr.state_dict().items()}
        asr = asr.to(cuda)
        audio, audio_len = torch.randn(4, 200, 64), torch.tensor([200, 150, 180, 99])
        text, text_len = torch.randint(1, 29, (4, 20)), torch.tensor([20, 12, 17, 9])
        if precision == "fp32":
            asr.eval()
            with torch.no_grad():
                got, got_len = asr(audio.to(cuda), audio_len.to(cuda))
                want, want_len = cnn.audio_to_align_text_forward(audio, audio_len, state, enc, 2, 512)
            assert torch.equal(got_len, want_len)
            assert rel_err(got, want) < 2e-4
            asr.train()
        asr.batch_augment.forward = lambda a, l: (a, l)
        opt = asr.configure_optimizers()
        batch = ((audio.to(cuda), audio_len.to(cuda)), (text.to(cuda), text_len.to(cuda)))
        losses = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss = asr.training_step(batch)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(l == l for l in losses) and losses[-1] < losses[0]

        tts = AlignTextToAudio(vocab_size=29, logspc_size=257, codeap_size=1, encoder_num_layers=2, encoder_hidden_size=512,
                               decoder_settings=dec).to(cuda)
        aligntext, aligntext_len = torch.randint(0, 29, (4, 60)).to(cuda), torch.tensor([60, 41, 55, 33]).to(cuda)
        f0, f0_len = (torch.rand(4, 119) * 200).to(cuda), torch.tensor([119, 81, 109, 65]).to(cuda)
        logspc, codeap = (torch.randn(4, 119, 257) - 6).to(cuda), (torch.randn(4, 119, 1) * 0.5 - 0.3).to(cuda)
        with torch.no_grad():
            tts.norm.f0_mean.fill_(100.0), tts.norm.f0_std.fill_(50.0), tts.norm.logspc_mean.fill_(-6.0)
        opt = tts.configure_optimizers()
        batch = ((f0, f0_len, logspc, codeap), (aligntext, aligntext_len))
        losses = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss = tts.training_step(batch)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(l == l for l in losses) and losses[-1] < losses[0]
        tts.eval()
        with torch.no_grad():
            f0p, logspcp, codeapp = tts.predict(aligntext, aligntext_len)
        assert f0p.shape == (4, 119) and logspcp.shape == (4, 119, 257) and codeapp.shape == (4, 119, 1)
    finally:
        F_.set_matmul_precision("fp32")
