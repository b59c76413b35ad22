#!/usr/bin/env python3
"""Golden vectors from the two THIRD-PARTY libraries the reference's feature pipeline leans on -- run on any machine that has them:

    pip install pyworld==0.3.2 torchaudio==0.13.1 torch==1.13.1 numpy        # the reference's pins: poetry.lock:1482-1484, 1796-1798, 1755-1757
    python tests/golden/make_thirdparty_vectors.py                           # -> tests/golden/thirdparty_world.npz, thirdparty_mel.npz

This script imports pyworld, torchaudio, torch and numpy ONLY -- not the reference, not this repository -- and restates nothing: it
makes exactly the CALLS the reference makes (voice100/vocoder.py:66-73 analysis, :99-101 synthesis; voice100/data_modules.py:276-291
log-mel) on fixed inputs and stores inputs and outputs.  Neither library is in the build image (no wheel, no network), so the
files do not exist in the tree yet; tests/test_thirdparty_pins.py (CPU: the oracle) and tests/test_gpu_thirdparty_pins.py (the
device kernels) compare against them as soon as they do, and skip with that reason until then.  With them in place the oracle rows
"A13 log-mel" and "f4 WORLD" stop being "parity unpinned".

Inputs (all 16 kHz mono, float64 in [-1, 1)):
  * the three reference docs samples held in tests/golden/world_ref_samples.npz (en1 whole, ja1_head / en2_head 1.5 s: real
    WORLD-synthesised speech, two of them starting unvoiced) when that file is beside this script;
  * two synthetic signals generated below: `chirp` (a harmonic tone gliding 110 -> 220 Hz behind 0.3 s of EXACT digital silence and
    followed by 0.2 s of it: what DIO does in silence is the one place the device and the oracle deliberately add a term, see
    oracle/world_analysis.py DIO_DITHER) and `mix` (voiced / unvoiced alternation over a noise floor, wandering F0 90-300 Hz).
Stored per signal <name>: x_<name>, and from pyworld: f0_, tpos_, sp_, ap_, codeap_, decoded ap (decode_aperiodicity of codeap) dap_,
y_ (synthesize of f0 / sp / decoded ap, frame_period 10): the calls of vocoder.py with its defaults f0_floor 80, f0_ceil 400,
frame_period 10, fft_size 512.  thirdparty_mel.npz: waveforms of 1 s and 10 s and torch.log(MelSpectrogram(...)(w).T + 1e-6).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FS, FRAME_PERIOD, N_FFT = 16000, 10.0, 512


def synthetic():
    rng = np.random.RandomState(20260505)
    n = int(1.6 * FS)
    t = np.arange(n) / FS
    f = 110.0 * 2.0 ** (t / t[-1])
    ph = 2 * np.pi * np.cumsum(f) / FS
    chirp = sum(np.sin(k * ph) / k for k in range(1, 9)) * 0.2
    chirp = np.concatenate([np.zeros(int(0.3 * FS)), chirp, np.zeros(int(0.2 * FS))])
    n = int(2.5 * FS)
    t = np.arange(n) / FS
    f0 = 180.0 + 90.0 * np.sin(2 * np.pi * 0.7 * t) + 30.0 * np.sin(2 * np.pi * 2.3 * t)
    ph = 2 * np.pi * np.cumsum(f0) / FS
    voiced = sum(np.sin(k * ph) * (0.6 ** k) for k in range(1, 12))
    gate = (np.sin(2 * np.pi * 1.1 * t) > -0.2).astype(np.float64)
    gate = np.convolve(gate, np.hanning(321) / np.hanning(321).sum(), mode="same")
    mix = 0.25 * voiced * gate + 0.02 * rng.randn(n) * (1.2 - gate)
    return {"chirp": chirp.astype(np.float64), "mix": mix.astype(np.float64)}


def world_vectors():
    import pyworld
    sig = synthetic()
    ref = os.path.join(HERE, "world_ref_samples.npz")
    if os.path.exists(ref):
        z = np.load(ref)
        for k in ("en1", "ja1_head", "en2_head"):
            sig[k] = z[k].astype(np.float64) / 32768.0
    out = {"pyworld_version": np.array(getattr(pyworld, "__version__", "unknown")), "names": np.array(sorted(sig))}
    for name, x in sig.items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        f0, tpos = pyworld.dio(x, FS, f0_floor=80.0, f0_ceil=400.0, frame_period=FRAME_PERIOD)           # vocoder.py:67-69
        sp = pyworld.cheaptrick(x, f0, tpos, FS, fft_size=N_FFT)                                          # :70
        ap = pyworld.d4c(x, f0, tpos, FS, fft_size=N_FFT)                                                 # :72
        codeap = pyworld.code_aperiodicity(ap, FS)                                                        # :73
        dap = pyworld.decode_aperiodicity(np.ascontiguousarray(codeap), FS, N_FFT)                        # :100
        y = pyworld.synthesize(np.ascontiguousarray(f0), np.ascontiguousarray(sp), dap, FS, frame_period=FRAME_PERIOD)   # :101
        out.update({f"x_{name}": x, f"f0_{name}": f0, f"tpos_{name}": tpos, f"sp_{name}": sp, f"ap_{name}": ap,
                    f"codeap_{name}": codeap, f"dap_{name}": dap, f"y_{name}": y})
    np.savez_compressed(os.path.join(HERE, "thirdparty_world.npz"), **out)
    print("wrote thirdparty_world.npz:", ", ".join(sorted(sig)))


def mel_vectors():
    import torch
    import torchaudio
    from torchaudio.transforms import MelSpectrogram
    g = torch.Generator().manual_seed(20260505)
    out = {"torchaudio_version": np.array(torchaudio.__version__), "torch_version": np.array(torch.__version__)}
    mel = MelSpectrogram(sample_rate=FS, n_fft=512, win_length=400, hop_length=160, n_mels=64)              # data_modules.py:276-281
    for name, secs in (("1s", 1.0), ("10s", 10.0)):
        n = int(secs * FS)
        t = torch.arange(n, dtype=torch.float32) / FS
        w = 0.3 * torch.sin(2 * np.pi * (200.0 + 150.0 * t / max(secs, 1.0)) * t) + 0.05 * torch.randn(n, generator=g)
        audio = torch.log(mel(w).T + 1e-6)                                                                  # :290-291
        out[f"w_{name}"] = w.numpy()
        out[f"logmel_{name}"] = audio.numpy()
    np.savez_compressed(os.path.join(HERE, "thirdparty_mel.npz"), **out)
    print("wrote thirdparty_mel.npz")


if __name__ == "__main__":
    what = sys.argv[1:] or ["world", "mel"]
    if "world" in what:
        world_vectors()
    if "mel" in what:
        mel_vectors()
