"""Log-mel batch augmentation on the GPU (drop-in for voice100/audio.py:17-108).

The seven randomly gated ops of the reference are evaluated by ONE fused HIP
kernel (csrc/augment.hip).  The random decisions are drawn here on the host with
the same `random` / `torch.rand` calls, in the same order, as the reference, so
a run seeded like the reference takes the same branches.
"""
import ctypes
import math
import random
from typing import Tuple

import torch
from torch import nn

from . import _native as N
from . import functional as F_

__all__ = ["BatchSpectrogramAugumentation"]

SPECTROGRAM_AUGUMENT_RATE = 0.2
LOG_OFFSET = 1e-6


class AugmentDecisions:
    """The numbers one forward() drew; tests inject a pre-built instance."""

    def __init__(self):
        self.stretch_rate = 0
        self.pitch_rate = 0.0
        self.amp = 0.0
        self.tmask = []            # [(t, hw, a)]
        self.fmask = None          # (t, hw, a)
        self.noise = None          # (low, high, std, uniform tensor or None)
        self.mix = False


class BatchSpectrogramAugumentation(nn.Module):
    """Augment mel-spectrogram data (class name as spelled by the reference)."""

    def __init__(self, do_timestretch=True, log_offset=LOG_OFFSET):
        super().__init__()
        self.do_timestretch = do_timestretch
        self.log_offset = log_offset
        self.blank_audio = math.log(log_offset)
        self.emit_transposed = False      # AudioToTextCTC turns this on: its forward() opens with the transpose

    def draw(self, audio: torch.Tensor) -> AugmentDecisions:
        """Consume `random` exactly as audio.py:34-49 does (shape-dependent draws included)."""
        d = AugmentDecisions()
        T, F = audio.shape[1], audio.shape[2]
        if self.do_timestretch and random.random() < SPECTROGRAM_AUGUMENT_RATE:
            d.stretch_rate = random.randrange(50, 150)
            if getattr(self, "_diag_rate", None):          # bench.py --diag-stretch-rate (diagnostics only): every stretch uses this rate
                d.stretch_rate = int(self._diag_rate)
            T = T * d.stretch_rate // 100
        if random.random() < SPECTROGRAM_AUGUMENT_RATE:
            d.pitch_rate = 1.0 + random.random() * 0.2
        if random.random() < SPECTROGRAM_AUGUMENT_RATE:
            d.amp = 1.0 + random.random() * 3.0
        if random.random() < SPECTROGRAM_AUGUMENT_RATE:
            for _ in range(random.randint(1, 3)):
                t = random.randrange(0, T)
                hw = random.randint(1, 3)
                d.tmask.append((t, hw, random.uniform(-self.blank_audio, -5)))
        if random.random() < SPECTROGRAM_AUGUMENT_RATE:
            t = random.randrange(0, F)
            hw = random.randint(1, 10)
            d.fmask = (t, hw, random.uniform(-self.blank_audio, -5))
        if random.random() < SPECTROGRAM_AUGUMENT_RATE:
            low = -5.0 + 5.0 * random.random()
            high = -5.0 + 5.0 * random.random()
            std = 5.0 * random.random()
            d.noise = (low, high, std, None)
        d.mix = random.random() < SPECTROGRAM_AUGUMENT_RATE
        return d

    @torch.no_grad()
    def forward(self, audio: torch.Tensor, audio_len: torch.Tensor, decisions: AugmentDecisions = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        assert len(audio.shape) == 3
        assert audio.dtype == torch.float32
        if not audio.is_cuda:
            raise RuntimeError("BatchSpectrogramAugumentation: voice100_amd runs on the GPU only")
        d = decisions if decisions is not None else self.draw(audio)
        return self.apply(audio, audio_len, d)

    def apply(self, audio, audio_len, d: AugmentDecisions):
        audio = audio.contiguous()
        B, Tin, F = audio.shape
        Tout = Tin
        if d.stretch_rate:
            Tout = Tin * d.stretch_rate // 100
        if Tout <= 0:
            raise RuntimeError("timestretch produced an empty batch")
        # the kernel takes the lengths BEFORE the stretch and derives len * rate // 100 (audio.py:58) itself; it also writes them
        # and the encoder's (len + 1) // 2 as side outputs, so the step spends no integer tensor ops on either
        len_raw = audio_len.to(device=audio.device, dtype=torch.int32).contiguous()
        len_pair = torch.empty((2, (B + 3) & ~3), dtype=torch.int32, device=audio.device)[:, :B]     # (rows 16-byte aligned)
        n = len(d.tmask)
        tm_s, tm_e, tm_a = (ctypes.c_int * 3)(), (ctypes.c_int * 3)(), (ctypes.c_float * 3)()
        for i, (t, hw, a) in enumerate(d.tmask):
            s, e, _ = slice(int(t - hw), int(t + hw)).indices(Tout)      # python slice semantics, audio.py:76-79
            tm_s[i], tm_e[i], tm_a[i] = s, max(e, s), a
        fm_on, fm_s, fm_e, fm_a = 0, 0, 0, 0.0
        if d.fmask is not None:
            t, hw, a = d.fmask
            s, e, _ = slice(int(t - hw), int(t + hw)).indices(F)
            fm_on, fm_s, fm_e, fm_a = 1, s, max(e, s), a
        uniform = None
        low = high = std = 0.0
        if d.noise is not None:
            low, high, std, uniform = d.noise
            if uniform is None:
                uniform = torch.rand((B, Tout, F), device=audio.device)
            uniform = uniform.to(audio.device).contiguous()
        out = torch.empty((B, Tout, F), dtype=torch.float32, device=audio.device)
        # the model that consumes the batch as [B, F, T] (asr.py:111) asks for that layout from the same pass (emit_transposed)
        out_t = torch.empty((B, F, Tout), dtype=torch.float32, device=audio.device) if (self.emit_transposed and F == 64) else None
        N.call("v100_augment_fused_len_t", audio, len_raw, len_pair[0], len_pair[1], uniform, out, out_t, B, Tin, Tout, F,
               int(d.stretch_rate), float(d.pitch_rate), float(d.amp), n, tm_s, tm_e, tm_a, fm_on, fm_s, fm_e, float(fm_a),
               int(d.noise is not None), float(low), float(high), float(std), int(d.mix), float(self.log_offset))
        if audio_len.device != audio.device:      # host lengths stay host lengths, computed there as the reference does (no sync)
            out_len = torch.div(audio_len * d.stretch_rate, 100, rounding_mode="trunc") if d.stretch_rate else audio_len.clone()
        else:
            out_len = len_pair[0] if audio_len.dtype == torch.int32 else len_pair[0].to(audio_len.dtype)
        if out_len.device == len_pair[1].device:  # host lengths get no device tag: output_length() then computes on the host like the reference
            F_.tag_half_length(out_len, len_pair[1])
        if out_t is not None:
            F_.tag_transposed(out, out_t)
        return out, out_len
