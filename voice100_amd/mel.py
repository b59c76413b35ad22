"""Log-mel front-end on the GPU (K7).  Drop-in arithmetic for MelSpectrogramAudioTransform
(voice100/data_modules.py:262-292): torchaudio-0.13.1 `MelSpectrogram(sample_rate=16000, n_fft=512,
win_length=400, hop_length=160, n_mels=64)` defaults (center, reflect pad, periodic Hann zero-padded to
n_fft, power 2, HTK mel scale, no norm) followed by log(mel.T + 1e-6).

The framed real DFT is a dense [2*257 x 400] x [400 x T] product (window folded into the basis) and the
filterbank a [64 x 257] x [257 x T] one; both run on the exact-fp32 MFMA GEMM kernel, with three small
HBM-bound kernels around them (framing, power, log+transpose).  File decoding / resampling stay on the
host (torchaudio), as in the reference.  PARITY UNPINNED against torchaudio itself (absent here): checked
against oracle/mel.py and torch.stft.
"""
import math

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import functional as F_

LOG_OFFSET = 1e-6
MELSPEC_DIM = 64


def _melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(np.float32(m_min), np.float32(m_max), n_mels + 2, dtype=np.float32)
    f_pts = (700.0 * (10.0 ** (m_pts / np.float32(2595.0)) - 1.0)).astype(np.float32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)        # [n_freqs, n_mels]


class MelSpectrogramAudioTransform(nn.Module):
    def __init__(self, sample_rate: int = 16000, n_fft: int = 512, win_length: int = 400, hop_length: int = 160,
                 n_mels: int = MELSPEC_DIM, log_offset: float = LOG_OFFSET) -> None:
        super().__init__()
        self.log_offset = log_offset
        self.sample_rate = sample_rate
        self.n_mels = n_mels
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        nf = n_fft // 2 + 1
        n = np.arange(win_length, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)              # periodic Hann
        left = (n_fft - win_length) // 2
        ang = 2.0 * np.pi * np.arange(nf, dtype=np.float64)[:, None] * (n[None, :] + left) / n_fft
        basis = np.concatenate([np.cos(ang) * window[None, :], -np.sin(ang) * window[None, :]], axis=0)
        self.register_buffer("dft_basis", torch.from_numpy(basis.astype(np.float32)), persistent=False)      # [2*nf, win]
        fb = _melscale_fbanks(nf, 0.0, sample_rate / 2.0, n_mels, sample_rate)
        self.register_buffer("mel_fb_t", torch.from_numpy(np.ascontiguousarray(fb.T)), persistent=False)      # [n_mels, nf]
        # tables of the one-launch kernel (csrc/mel.hip): n_fft = 512, <= 64 filters, each a contiguous run of <= 32 bins
        self.fused = False
        if n_fft == 512 and n_mels <= 64:
            starts, counts, wts, ok = np.zeros(n_mels, np.int32), np.zeros(n_mels, np.int32), np.zeros((n_mels, 32), np.float32), True
            for m in range(n_mels):
                nz = np.nonzero(fb[:, m])[0]
                if nz.size == 0:
                    continue
                lo, hi = int(nz[0]), int(nz[-1]) + 1
                if hi - lo > 32:
                    ok = False
                    break
                starts[m], counts[m] = lo, hi - lo
                wts[m, :hi - lo] = fb[lo:hi, m]                     # zeros inside the run (none for triangles) stay zero weights
            if ok:
                win512 = np.zeros(n_fft, dtype=np.float64)
                win512[left:left + win_length] = window
                k256, k512 = np.arange(256, dtype=np.float64), np.arange(257, dtype=np.float64)
                tw = lambda k, n_: np.stack([np.cos(2.0 * np.pi * k / n_), -np.sin(2.0 * np.pi * k / n_)], axis=1).astype(np.float32)
                self.register_buffer("_win512", torch.from_numpy(win512.astype(np.float32)), persistent=False)
                self.register_buffer("_tw256", torch.from_numpy(tw(k256, 256.0)), persistent=False)
                self.register_buffer("_tw512", torch.from_numpy(tw(k512, 512.0)), persistent=False)
                self.register_buffer("_mel_start", torch.from_numpy(starts), persistent=False)
                self.register_buffer("_mel_count", torch.from_numpy(counts), persistent=False)
                self.register_buffer("_mel_w", torch.from_numpy(wts), persistent=False)
                self.fused = True

    @property
    def audio_size(self) -> int:
        return self.n_mels

    def num_frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop_length

    @torch.no_grad()
    def transform(self, waveform: torch.Tensor, fused: bool = None) -> torch.Tensor:
        """waveform [N] or [B, N] fp32 on the GPU -> log-mel [T, n_mels] or [B, T, n_mels].  One launch (csrc/mel.hip: FFT per wave)
        for the reference's configuration; `fused=False` (or any other n_fft / filter count) takes the framing + DFT-GEMM +
        filterbank-GEMM kernels."""
        squeeze = waveform.dim() == 1
        x = waveform[None] if squeeze else waveform
        F_._check(x, "MelSpectrogramAudioTransform")
        x = x.contiguous()
        B, n = x.shape
        T = self.num_frames(n)
        nf = self.n_fft // 2 + 1
        if self.fused if fused is None else (fused and self.fused):
            out = torch.empty((B, T, self.n_mels), dtype=torch.float32, device=x.device)
            N.call("v100_log_mel_fused", x, out, self._win512, self._tw256, self._tw512, self._mel_start, self._mel_count, self._mel_w,
                   B, n, T, self.hop_length, self.n_fft, self.n_mels, float(self.log_offset))
            return out[0] if squeeze else out
        frames = torch.empty((B, self.win_length, T), dtype=torch.float32, device=x.device)
        N.call("v100_stft_frames", x, frames, B, n, T, self.hop_length, self.win_length, self.n_fft)
        spec = torch.empty((B, 2 * nf, T), dtype=torch.float32, device=x.device)
        F_._pw_gemm(self.dft_basis, None, frames, spec, 2 * nf, self.win_length, T, B, False)     # exact-fp32 MFMA
        power = torch.empty((B, nf, T), dtype=torch.float32, device=x.device)
        N.call("v100_power_spectrum", spec, power, B, nf, T)
        mel = torch.empty((B, self.n_mels, T), dtype=torch.float32, device=x.device)
        F_._pw_gemm(self.mel_fb_t, None, power, mel, self.n_mels, nf, T, B, False)
        out = torch.empty((B, T, self.n_mels), dtype=torch.float32, device=x.device)
        N.call("v100_log_transpose", mel, out, B, self.n_mels, T, float(self.log_offset))
        return out[0] if squeeze else out

    def forward(self, audio) -> torch.Tensor:
        """A path (as in the reference: decode + resample on the host with torchaudio, then the GPU transform)
        or an already-loaded waveform tensor."""
        if isinstance(audio, torch.Tensor):
            return self.transform(audio)
        try:
            import torchaudio
        except ImportError as e:
            raise RuntimeError("loading audio files needs torchaudio (not installed); pass a waveform tensor") from e
        waveform, sr = torchaudio.load(audio)
        waveform = torchaudio.functional.resample(waveform[0], sr, self.sample_rate)
        return self.transform(waveform.to(self.dft_basis.device))
