#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run in the build container only (needs /root/reference; the GPU box has none):

    python tests/golden/make_golden.py

The reference (kaiidams/voice100 v1.6.0) imports pytorch_lightning / pyworld,
which are absent here; they are stubbed in-process with empty modules (the
hot path uses none of their behaviour: LightningModule is only a base class).
Only inputs, parameters and outputs (data) are written -- no reference source.
"""
import os
import random
import sys
import types

import numpy as np
import torch

REF = os.environ.get("VOICE100_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    from torch import nn

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            self.hparams = types.SimpleNamespace()

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pyworld"] = types.ModuleType("pyworld")


def _np(d):
    return {k: (v.detach().cpu().numpy().copy() if isinstance(v, torch.Tensor) else np.array(v)) for k, v in d.items()}


def _state(module, prefix="state/"):
    return {prefix + k: v for k, v in module.state_dict().items()}


def _randomize_bn(module, gen):
    """Give BN layers non-trivial affine params / running stats so eval parity is a real test."""
    from torch import nn
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.3)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def gen_ir_blocks():
    from voice100.models.asr import InvertedResidual
    out = {}
    # (cin, cout, k, stride, residual, B, T)
    cfgs = [
        (8, 16, 11, 2, False, 2, 61),
        (8, 8, 19, 1, True, 2, 48),
        (16, 16, 83, 1, True, 2, 96),
        (8, 16, 51, 1, False, 3, 40),
        (16, 16, 5, 1, True, 2, 33),
        (8, 8, 7, 1, True, 1, 17),
    ]
    gen = torch.Generator().manual_seed(1234)
    for n, (cin, cout, k, s, res, B, T) in enumerate(cfgs):
        torch.manual_seed(1234 + n)
        m = InvertedResidual(cin, cout, kernel_size=k, stride=s, use_residual=res)
        _randomize_bn(m, gen)
        x = torch.randn(B, cin, T, generator=gen)
        pre = f"b{n}/"
        out[pre + "cfg"] = np.array([cin, cout, k, s, int(res), B, T], dtype=np.int64)
        out[pre + "x"] = x.numpy()
        for key, v in m.state_dict().items():
            out[pre + "state/" + key] = v.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[pre + "y_eval"] = m(x).numpy()
        m.train()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out[pre + "y_train"] = y.detach().numpy()
        out[pre + "gy"] = gy.numpy()
        out[pre + "gx"] = xg.grad.numpy()
        for key, p in m.named_parameters():
            out[pre + "grad/" + key] = p.grad.numpy()
        for key, v in m.state_dict().items():
            if "running" in key or "num_batches" in key:
                out[pre + "after/" + key] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "ir_blocks.npz"), **out)


def gen_asr_tiny():
    from voice100.models.asr import AudioToTextCTC
    torch.manual_seed(1234)
    m = AudioToTextCTC(audio_size=64, embed_size=32, vocab_size=29, hidden_size=32,
                       learning_rate=1e-3, weight_decay=4e-5)
    gen = torch.Generator().manual_seed(99)
    _randomize_bn(m, gen)
    m.decoder.layers[0].p = 0.0          # dropout off: train-mode parity must be reproducible
    B, T = 3, 120
    audio = torch.randn(B, T, 64, generator=gen) * 2 - 4
    audio_len = torch.tensor([120, 97, 64], dtype=torch.int32)
    text = torch.randint(1, 29, (B, 12), generator=gen)
    text_len = torch.tensor([12, 9, 5], dtype=torch.int32)
    out = {"audio": audio.numpy(), "audio_len": audio_len.numpy(), "text": text.numpy(), "text_len": text_len.numpy()}
    out.update(_np(_state(m)))
    m.eval()
    with torch.no_grad():
        logits = m(audio)
    out["logits_eval"] = logits.numpy()
    out["argmax_eval"] = logits.argmax(-1).numpy()
    out["output_length"] = m.output_length(audio_len).numpy()
    m.train()
    class _NoAugment(torch.nn.Module):      # augmentation is tested separately with injected decisions
        def forward(self, a, l):
            return a, l
    m.batch_augment = _NoAugment()
    loss = m._calc_batch_loss(((audio, audio_len), (text, text_len)))
    loss.backward()
    out["loss_train"] = loss.detach().numpy()
    for key, p in m.named_parameters():
        out["grad/" + key] = p.grad.numpy()
    for key, v in m.state_dict().items():
        if "running" in key or "num_batches" in key:
            out["after/" + key] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "asr_tiny.npz"), **out)


def gen_asr_c1():
    """BASELINE config 1: full-size asr_en_base eval forward, B=2, torch.rand(2,256,64).
    Weights are NOT stored (46 MB): they are the torch default init under
    torch.manual_seed(1234) with the reference's construction order; a checksum of
    them is stored so a consumer can tell whether it reproduced the same init."""
    from voice100.models.asr import AudioToTextCTC
    torch.manual_seed(1234)
    m = AudioToTextCTC(audio_size=64, embed_size=512, vocab_size=29, hidden_size=512,
                       learning_rate=1e-3, weight_decay=4e-5)
    audio = torch.rand(2, 256, 64)
    m.eval()
    with torch.no_grad():
        logits = m(audio)
    sd = m.state_dict()
    wsum = np.array([float(v.double().sum()) for k, v in sd.items() if v.dtype.is_floating_point])
    wabs = np.array([float(v.double().abs().sum()) for k, v in sd.items() if v.dtype.is_floating_point])
    out = {
        "audio": audio.numpy(),
        "logits": logits.numpy(),               # [2,128,29] = 30 KB
        "argmax": logits.argmax(-1).numpy(),
        "weight_sums": wsum, "weight_abs_sums": wabs,
        "n_params": np.array(sum(p.numel() for p in m.parameters())),
    }
    np.savez_compressed(os.path.join(OUT, "asr_c1.npz"), **out)


def gen_tts_tiny():
    from voice100.models.tts import AlignTextToAudioModel, TextToAlignTextModel, VoiceDecoder
    gen = torch.Generator().manual_seed(7)
    for use_mcep in (False, True):
        torch.manual_seed(1234)
        m = AlignTextToAudioModel(vocab_size=29, hidden_size=32, learning_rate=1e-3, use_mcep=use_mcep)
        _randomize_bn(m, gen)
        with torch.no_grad():
            m.norm.f0_mean.fill_(120.0); m.norm.f0_std.fill_(35.0)
            m.norm.logspc_mean.copy_(torch.randn(m.norm.logspc_mean.shape, generator=gen) - 5)
            m.norm.logspc_std.copy_(torch.rand(m.norm.logspc_std.shape, generator=gen) + 0.5)
            m.norm.codeap_mean.fill_(-2.0); m.norm.codeap_std.fill_(1.5)
        B, L = 2, 40
        aligntext = torch.randint(0, 29, (B, L), generator=gen)
        out = {"aligntext": aligntext.numpy()}
        out.update(_np(_state(m)))
        m.eval()
        with torch.no_grad():
            fwd = m(aligntext)
            pred = m.predict(aligntext)
        for name, v in zip(("hasf0_logits", "f0_hat", "logspc_hat", "codeap_hat"), fwd):
            out["fwd/" + name] = v.numpy()
        for name, v in zip(("f0", "logspc", "codeap"), pred):
            out["predict/" + name] = v.numpy()
        # training loss + grads with synthetic WORLD targets (collate layout data_modules.py:458-474)
        Tw = 2 * L + 3
        f0 = torch.rand(B, Tw, generator=gen) * 200
        f0 = torch.where(f0 < 60, torch.zeros(()), f0)
        f0_len = torch.tensor([Tw, Tw - 11], dtype=torch.int32)
        logspc = torch.randn(B, Tw, m.logspc_size, generator=gen) - 5
        codeap = torch.randn(B, Tw, 1, generator=gen) - 2
        aligntext_len = torch.tensor([L, L - 5], dtype=torch.int32)
        m.train()
        losses = m._calc_batch_loss(((f0, f0_len, logspc, codeap), (aligntext, aligntext_len)))
        sum(losses).backward()
        out["target/f0"] = f0.numpy(); out["target/f0_len"] = f0_len.numpy()
        out["target/logspc"] = logspc.numpy(); out["target/codeap"] = codeap.numpy()
        out["losses_train"] = np.array([float(v) for v in losses], dtype=np.float32)
        for key, p in m.named_parameters():
            if p.grad is not None:
                out["grad/" + key] = p.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"tts_tiny_{'mcep' if use_mcep else 'logspc'}.npz"), **out)

    # TextToAlignTextModel + align()
    torch.manual_seed(1234)
    m = TextToAlignTextModel(vocab_size=29, hidden_size=32, learning_rate=1e-3)
    _randomize_bn(m, gen)
    text = torch.randint(1, 29, (2, 24), generator=gen)
    out = {"text": text.numpy()}
    out.update(_np(_state(m)))
    m.eval()
    with torch.no_grad():
        out["pred_eval"] = m(text).numpy()
    text_len = torch.tensor([24, 17], dtype=torch.int32)
    align = torch.randint(0, 6, (2, 49), generator=gen)
    align_len = torch.tensor([49, 35], dtype=torch.int32)
    m.train()
    loss = m._calc_batch_loss(((text, text_len), (align, align_len)))
    loss.backward()
    out["align"] = align.numpy(); out["text_len"] = text_len.numpy(); out["align_len"] = align_len.numpy()
    out["loss_train"] = loss.detach().numpy()
    for key, p in m.named_parameters():
        out["grad/" + key] = p.grad.numpy()
    # align(): integer expansion, float and integer (gap,len) pairs incl. .5 ties
    cases = [
        (torch.tensor([3, 7, 7, 2, 11]), torch.tensor([[0.0, 2.0], [1.5, 1.0], [0.5, 0.5], [0.0, 0.2], [2.5, 3.5]])),
        (torch.tensor([5, 1, 9]), torch.tensor([[1, 2], [0, 3], [2, 1]])),
        (torch.tensor([4, 4, 8, 15]), torch.tensor([[0.49, 0.02], [0.0, 0.0], [0.3, 4.2], [1.7, 0.8]])),
    ]
    for n, (t, a) in enumerate(cases):
        out[f"align_case{n}/text"] = t.numpy(); out[f"align_case{n}/align"] = a.numpy()
        out[f"align_case{n}/aligntext"] = m.align(t, a, head=5, tail=5).numpy()
    np.savez_compressed(os.path.join(OUT, "align_tiny.npz"), **out)

    # isolated ConvTranspose1d (tts.py:22) and a tiny VoiceDecoder
    torch.manual_seed(5)
    ct = torch.nn.ConvTranspose1d(16, 8, kernel_size=5, padding=2, stride=2)
    x = torch.randn(2, 16, 21, generator=gen, requires_grad=True)
    y = ct(x)
    gy = torch.randn(y.shape, generator=gen)
    (y * gy).sum().backward()
    np.savez_compressed(os.path.join(OUT, "convtranspose.npz"), x=x.detach().numpy(), weight=ct.weight.detach().numpy(),
                        bias=ct.bias.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), gx=x.grad.numpy(),
                        gw=ct.weight.grad.numpy(), gb=ct.bias.grad.numpy())


def gen_augment():
    from voice100.audio import BatchSpectrogramAugumentation
    aug = BatchSpectrogramAugumentation()
    gen = torch.Generator().manual_seed(3)
    B, T = 4, 50
    audio = torch.randn(B, T, 64, generator=gen) * 2 - 4
    audio_len = torch.tensor([50, 41, 33, 7], dtype=torch.int32)
    out = {"audio": audio.numpy(), "audio_len": audio_len.numpy()}

    def with_seed(seed, fn, *a):
        random.seed(seed)
        torch.manual_seed(seed)
        st = random.getstate()
        r = fn(*a)
        random.setstate(st)
        return r

    # timestretch: record the drawn rate
    for n, seed in enumerate((1, 2, 3)):
        random.seed(seed); rate = random.randrange(50, 150)
        random.seed(seed); a, l = aug.timestretch(audio, audio_len)
        out[f"timestretch{n}/rate"] = np.array(rate); out[f"timestretch{n}/audio"] = a.numpy(); out[f"timestretch{n}/len"] = l.numpy()
    for n, seed in enumerate((4, 5)):
        random.seed(seed); rate = 1.0 + random.random() * 0.2
        random.seed(seed); a = aug.pitchshift(audio)
        out[f"pitchshift{n}/rate"] = np.array(rate, dtype=np.float64); out[f"pitchshift{n}/audio"] = a.numpy()
    random.seed(6); rate = 1.0 + random.random() * 3.0
    random.seed(6); out["ampshift/audio"] = aug.ampshift(audio).numpy(); out["ampshift/rate"] = np.array(rate)
    for n, seed in enumerate((7, 8, 21)):
        random.seed(seed)
        k = random.randint(1, 3); spans = []
        for _ in range(k):
            t = random.randrange(0, T); hw = random.randint(1, 3); a_ = random.uniform(-aug.blank_audio, -5)
            spans.append((t, hw, a_))
        random.seed(seed); a = aug.timemask(audio)
        out[f"timemask{n}/spans"] = np.array(spans, dtype=np.float64); out[f"timemask{n}/audio"] = a.numpy()
    for n, seed in enumerate((9, 10, 11)):
        random.seed(seed)
        t = random.randrange(0, 64); hw = random.randint(1, 10); a_ = random.uniform(-aug.blank_audio, -5)
        random.seed(seed); a = aug.freqmask(audio)
        out[f"freqmask{n}/params"] = np.array([t, hw, a_], dtype=np.float64); out[f"freqmask{n}/audio"] = a.numpy()
    random.seed(12)
    low = -5.0 + 5.0 * random.random(); high = -5.0 + 5.0 * random.random(); std = 5.0 * random.random()
    torch.manual_seed(12); u = torch.rand(audio.shape)
    random.seed(12); torch.manual_seed(12); a = aug.mixnoise(audio)
    out["mixnoise/params"] = np.array([low, high, std]); out["mixnoise/uniform"] = u.numpy(); out["mixnoise/audio"] = a.numpy()
    out["mixaudio/audio"] = aug.mixaudio(audio, audio_len).numpy()
    out["maskaudio/audio"] = aug.maskaudio(audio, audio_len).numpy()
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)


def gen_mcep_and_int():
    from voice100.vocoder import create_sp2mc_matrix, create_mc2sp_matrix
    from voice100.models.align import ctc_best_path
    from voice100.models._layers_v1 import generate_padding_mask
    from voice100.models.asr import ConvVoiceEncoder
    out = {
        "sp2mc_16k": create_sp2mc_matrix(512, 24, 0.410), "mc2sp_16k": create_mc2sp_matrix(512, 24, 0.410),
        "sp2mc_22k": create_sp2mc_matrix(1024, 34, 0.455), "mc2sp_22k": create_mc2sp_matrix(1024, 34, 0.455),
    }
    np.savez_compressed(os.path.join(OUT, "mcep.npz"), **out)

    out = {}
    lens = torch.arange(0, 40, dtype=torch.int32)
    enc = ConvVoiceEncoder(8, 8, 8)
    out["output_length/in"] = lens.numpy(); out["output_length/out"] = enc.output_length(lens).numpy()
    x = torch.zeros(4, 9); ln = torch.tensor([9, 0, 4, 12])
    out["padding_mask/len"] = ln.numpy(); out["padding_mask/mask"] = generate_padding_mask(x, ln).numpy()
    rng = np.random.RandomState(11)
    for n, (T, L, V) in enumerate(((20, 4, 6), (33, 7, 29), (15, 6, 10))):
        lp = torch.log_softmax(torch.from_numpy(rng.randn(T, V).astype(np.float32)), -1).numpy()
        labels = rng.randint(1, V, size=L).astype(np.int64)
        if n == 1:
            labels[2] = labels[1]     # repeated label
        score, path, best = ctc_best_path(lp, labels)
        out[f"ctc{n}/logits"] = lp; out[f"ctc{n}/labels"] = labels
        out[f"ctc{n}/score"] = np.array(score); out[f"ctc{n}/path"] = path; out[f"ctc{n}/best_labels"] = best
    np.savez_compressed(os.path.join(OUT, "int_tables.npz"), **out)


def gen_v2_blocks():
    """v2 conv stacks: get_conv_layers with the channel counts scaled down, the (k, stride,
    padding, transpose, bias) pattern of config/asr_en_base.yaml:16-18 and config/tts_en_base.yaml:20-23."""
    from voice100.models._layers_v2 import get_conv_layers
    out = {}
    gen = torch.Generator().manual_seed(4321)
    # (in_channels, settings, B, T)
    cfgs = [
        (24, [[40, False, 5, 2, 2, False], [40, False, 5, 1, 2, False]], 2, 51),        # ASR encoder pattern
        (48, [[40, False, 5, 1, 2, False], [40, True, 5, 2, 2, False], [40, False, 5, 1, 2, False]], 2, 23),   # TTS decoder
        (16, [[72, False, 5, 2, 2, True], [72, True, 5, 2, 2, True]], 3, 34),           # biases on, C not a multiple of 32
    ]
    for n, (cin, settings, B, T) in enumerate(cfgs):
        torch.manual_seed(4321 + n)
        m = get_conv_layers(cin, settings)
        from torch import nn
        for mod in m.modules():
            if isinstance(mod, nn.LayerNorm):
                with torch.no_grad():
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=gen) + 0.5)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.3)
        x = torch.randn(B, cin, T, generator=gen)
        pre = f"s{n}/"
        out[pre + "cin"] = np.array(cin, dtype=np.int64)
        out[pre + "settings"] = np.array([[int(v) for v in row] for row in settings], dtype=np.int64)
        out[pre + "x"] = x.numpy()
        for key, v in m.state_dict().items():
            out[pre + "state/" + key] = v.numpy().copy()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        out[pre + "y"] = y.detach().numpy()
        out[pre + "gy"] = gy.numpy()
        out[pre + "gx"] = xg.grad.numpy()
        for key, p in m.named_parameters():
            out[pre + "grad/" + key] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "v2_blocks.npz"), **out)


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    _install_stubs()
    torch.set_num_threads(8)
    gen_ir_blocks()
    gen_asr_tiny()
    gen_asr_c1()
    gen_tts_tiny()
    gen_augment()
    gen_mcep_and_int()
    gen_v2_blocks()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
