"""Log-mel front-end restatement.  TEST INFRASTRUCTURE -- PARITY UNPINNED.

Reference call site: voice100/data_modules.py:262-292
(`MelSpectrogram(sample_rate=16000, n_fft=512, win_length=400, hop_length=160,
n_mels=64)` then `log(mel.T + 1e-6)`).  The arithmetic lives in torchaudio
0.13.1 (poetry.lock:1796-1798), which is not in the reference tree nor in this
image, so this follows its published definition:
  Spectrogram: center=True, pad_mode="reflect", periodic Hann(win_length)
  zero-padded (centred) to n_fft, onesided, power=2, normalized=False;
  MelScale: f_min=0, f_max=sr/2, mel_scale="htk", norm=None, triangular
  filters built in float32 from linspace'd mel points.
No golden vector exists for it; tests cross-check it against torch.stft only.
"""
import numpy as np


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def melscale_fbanks(n_freqs=257, f_min=0.0, f_max=8000.0, n_mels=64, sample_rate=16000):
    """[n_freqs, n_mels] float32 triangles (torchaudio.functional.melscale_fbanks)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_pts = np.linspace(np.float32(hz_to_mel_htk(f_min)), np.float32(hz_to_mel_htk(f_max)),
                        n_mels + 2, dtype=np.float32)
    f_pts = mel_to_hz_htk(m_pts.astype(np.float32)).astype(np.float32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


def hann_window_padded(win_length=400, n_fft=512):
    """Periodic Hann of win_length, centred in n_fft zeros (torch.stft semantics)."""
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    out = np.zeros(n_fft, dtype=np.float64)
    left = (n_fft - win_length) // 2
    out[left:left + win_length] = w
    return out.astype(np.float32)


def num_frames(n_samples, hop_length=160):
    return 1 + n_samples // hop_length


def log_mel(waveform, sample_rate=16000, n_fft=512, win_length=400, hop_length=160,
            n_mels=64, log_offset=1e-6):
    """waveform [N] float32 -> [1 + N//hop, n_mels] float32 (data_modules.py:287-292)."""
    x = np.asarray(waveform, dtype=np.float32)
    pad = n_fft // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    t = num_frames(x.shape[0], hop_length)
    win = hann_window_padded(win_length, n_fft)
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(t)[:, None]
    frames = xp[idx] * win[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate)
    mel = power @ fb
    return np.log(mel + np.float32(log_offset)).astype(np.float32)
