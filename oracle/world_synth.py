"""WORLD synthesis restatement (float64 numpy).  TEST INFRASTRUCTURE -- PARITY PARTIALLY PINNED (no input/output vector; see below).

Reference call site: voice100/vocoder.py:89-102 --
    ap = pyworld.decode_aperiodicity(codeap, sample_rate, n_fft)
    waveform = pyworld.synthesize(f0, spc, ap, sample_rate, frame_period=frame_period)
The arithmetic lives in pyworld 0.3.2 (poetry.lock: pyworld 0.3.2, a Cython wrapper over M. Morise's C++ WORLD library), which is
NOT in the reference tree and not in this image, and no input/output pair of pyworld exists here.  What does exist are the reference's own
docs samples -- waveforms its TTS chain synthesised with pyworld -- and they pin part of this file (tests/golden/README.md,
tests/test_oracle_world_analysis.py): the output length; the time base's phase and the response placement (an unvoiced start pulses at the
500 Hz default from sample 30, the first response begins at sample 31: the reference's files lead with exactly 31 zeros, as does this
restatement); and the whole NOISE path -- WORLD's xorshift randn, burst per pulse, mean removal, minimum-phase colouring, fftshift and
overlap-add -- because a re-synthesis from an analysis of those files reproduces their unvoiced lead-in sample by sample (correlation
0.93 - 0.98 per 20-ms window; any other generator / alignment gives ~0).  The PERIODIC path (fractional-delay pulse, DC removal) and the
exact F0-to-pulse arithmetic beyond the default rate have no such witness.  The file restates the published algorithm --
  * M. Morise, F. Yokomori, K. Ozawa, "WORLD: a vocoder-based high-quality speech synthesis system for real-time applications",
    IEICE Trans. Inf. & Syst. E99-D(7), 2016 (synthesis: excitation pulses at the F0-derived instants, a minimum-phase response
    per pulse from the spectral envelope, periodic + aperiodic parts, overlap-add);
  * M. Morise, "D4C, a band-aperiodicity estimator for high-quality speech synthesis", Speech Communication 84, 2016 (the coded
    band aperiodicity in dB on a 3 kHz grid, linearly interpolated to the FFT bins);
  * the structure of the open-source implementation as its author documents it (synthesis.cpp: GetTimeBase /
    GetOneFrameSegment / GetPeriodicResponse / GetAperiodicResponse; codec.cpp: DecodeAperiodicity; common.cpp:
    GetMinimumPhaseSpectrum, interp1; matlabfunctions.cpp: randn as a sum of twelve xorshift128 uniforms) --
and is checked by PROPERTIES (tests/test_oracle_world.py): pitch and envelope of a constant-F0 resynthesis, noise level of unvoiced
frames, linearity in the spectral amplitude, time-shift consistency.  Constants: default F0 of unvoiced frames 500 Hz, safeguard
1e-12, aperiodicity clipped to [0.001, 0.999999999999], band interval 3 kHz (upper limit 15 kHz), unvoiced threshold -0.5 dB.

Every function takes / returns float64 numpy arrays.  `synthesize_parts` also returns the intermediate quantities the HIP kernels are
compared against stage by stage (pulse instants, per-pulse responses).
"""
import ctypes
import os

import numpy as np

K_DEFAULT_F0 = 500.0
K_SAFE_MIN = 1e-12
K_FREQ_INTERVAL = 3000.0
K_UPPER_LIMIT = 15000.0
_HERE = os.path.dirname(os.path.abspath(__file__))


# ---- randn(): sum of twelve xorshift128 uniforms minus six (WORLD matlabfunctions.cpp) ------------------------------------
def _randn_table_py(n):
    x, y, z, w = 123456789, 362436069, 521288629, 88675123
    M = 0xFFFFFFFF
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        tmp = 0
        for _ in range(12):
            t = (x ^ (x << 11)) & M
            x, y, z = y, z, w
            w = ((w ^ (w >> 19)) ^ (t ^ (t >> 8))) & M
            tmp = (tmp + (w >> 4)) & M
        out[i] = tmp / 268435456.0 - 6.0
    return out


_TABLE = np.empty(0)


def randn_table(n):
    """First n values of WORLD's randn() after randn_reseed() (every Synthesis call reseeds, so pulse i of ANY utterance draws the
    values [idx_i - idx_0, idx_{i+1} - idx_0) of this one fixed sequence).  C helper (oracle/_build/libworld_randn.so, `make -C
    oracle`) when built, pure Python otherwise."""
    global _TABLE
    if _TABLE.size < n:
        n_alloc = max(n, 2 * _TABLE.size)
        lib = os.path.join(_HERE, "_build", "libworld_randn.so")
        if os.path.exists(lib):
            f = ctypes.CDLL(lib).world_randn_fill
            f.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_long]
            buf = np.empty(n_alloc, dtype=np.float64)
            f(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n_alloc)
            _TABLE = buf
        else:
            _TABLE = _randn_table_py(n_alloc)
    return _TABLE[:n]


# ---- interp1 (WORLD common.cpp: histc + one linear segment per query, end segments extrapolate) ------------------------------
def interp1(x, y, xi):
    x, y, xi = np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(xi, np.float64)
    k = np.searchsorted(x, xi, side="right")            # x[k-1] <= xi < x[k]
    k = np.clip(k, 1, len(x) - 1)
    h = x[k] - x[k - 1]
    s = (xi - x[k - 1]) / h
    return y[k - 1] + s * (y[k] - y[k - 1])


# ---- DecodeAperiodicity (codec.cpp) -----------------------------------------------------------------------------------------
def number_of_aperiodicities(fs):
    return int(min(K_UPPER_LIMIT, fs / 2.0 - K_FREQ_INTERVAL) / K_FREQ_INTERVAL)


def decode_aperiodicity(coded_ap, fs, fft_size):
    """coded_ap [T, nb] (dB) -> aperiodicity [T, fft_size/2 + 1] in (0, 1): frames whose mean coded value exceeds -0.5 dB are
    unvoiced and keep 1 - 1e-12 everywhere; the others are interpolated linearly in dB over (0 Hz: -60 dB, 3 kHz bands, fs/2:
    -1e-12 dB) and converted with 10^(dB/20)."""
    coded_ap = np.asarray(coded_ap, np.float64)
    T, nb = coded_ap.shape
    if nb != number_of_aperiodicities(fs):
        raise ValueError("coded aperiodicity width does not match the sampling rate")
    ap = np.full((T, fft_size // 2 + 1), 1.0 - K_SAFE_MIN)
    freq = float(fs) / fft_size * np.arange(fft_size // 2 + 1)
    coarse_f = np.concatenate([np.arange(nb + 1) * K_FREQ_INTERVAL, [fs / 2.0]])
    for t in range(T):
        if coded_ap[t].sum() / nb > -0.5:
            continue
        coarse = np.concatenate([[-60.0], coded_ap[t], [-K_SAFE_MIN]])
        ap[t] = 10.0 ** (interp1(coarse_f, coarse, freq) / 20.0)
    return ap


# ---- time base: pulse instants from the F0 contour (synthesis.cpp GetTimeBase) ----------------------------------------------
def time_base(f0, fs, frame_period_ms, y_length, fft_size):
    f0 = np.asarray(f0, np.float64)
    T = len(f0)
    fp = frame_period_ms / 1000.0
    lowest_f0 = fs / fft_size + 1.0
    time_axis = np.arange(y_length) / float(fs)
    coarse_t = np.arange(T + 1) * fp
    coarse_f0 = np.where(f0 < lowest_f0, 0.0, f0)
    coarse_vuv = np.where(coarse_f0 == 0.0, 0.0, 1.0)
    coarse_f0 = np.concatenate([coarse_f0, [coarse_f0[-1] * 2 - coarse_f0[-2]]])
    coarse_vuv = np.concatenate([coarse_vuv, [coarse_vuv[-1] * 2 - coarse_vuv[-2]]])
    f0i = interp1(coarse_t, coarse_f0, time_axis)
    vuv = (interp1(coarse_t, coarse_vuv, time_axis) > 0.5).astype(np.float64)
    f0i = np.where(vuv == 0.0, K_DEFAULT_F0, f0i)
    total = np.cumsum(2.0 * np.pi * f0i / fs)           # sequential accumulation, as the C loop
    wrap = np.fmod(total, 2.0 * np.pi)
    jump = np.abs(wrap[1:] - wrap[:-1]) > np.pi
    idx = np.nonzero(jump)[0]
    y1 = wrap[idx] - 2.0 * np.pi
    y2 = wrap[idx + 1]
    shift = (-y1 / (y2 - y1)) / fs
    return idx.astype(np.int64), time_axis[idx], shift, vuv


# ---- minimum-phase spectrum from half a power-like log spectrum (common.cpp GetMinimumPhaseSpectrum) ------------------------
def minimum_phase(log_amp_half, fft_size):
    """log_amp_half [.., N/2+1] = log(amplitude) -> complex minimum-phase spectrum [.., N/2+1] = exp(FFT(fold(IFFT(mirror))))."""
    N = fft_size
    full = np.concatenate([log_amp_half, log_amp_half[..., -2:0:-1]], axis=-1)
    cep = np.fft.fft(full, axis=-1).real / N
    fold = np.zeros_like(cep)
    fold[..., 0] = cep[..., 0]
    fold[..., 1:N // 2] = 2.0 * cep[..., 1:N // 2]
    fold[..., N // 2] = cep[..., N // 2]
    m = np.fft.fft(fold, axis=-1)[..., :N // 2 + 1]
    return np.exp(m.real) * (np.cos(m.imag) + 1j * np.sin(m.imag))


def dc_remover(fft_size):
    i = np.arange(fft_size // 2)
    half = 0.5 - 0.5 * np.cos(2.0 * np.pi * (i + 1.0) / (1.0 + fft_size))
    w = np.concatenate([half, half[::-1]])
    return w / (2.0 * half.sum())


def _frame_mix(cur_t, fp, T):
    pos = cur_t / fp
    fl = np.minimum(T - 1, np.floor(pos).astype(np.int64))
    ce = np.minimum(T - 1, np.ceil(pos).astype(np.int64))
    return fl, ce, pos - fl


def synthesize_parts(f0, sp, ap, fs, frame_period=5.0):
    """Returns dict(y, idx, shift, vuv, noise_size, response [n_pulses, fft_size]) -- WORLD Synthesis() pulse by pulse."""
    f0 = np.asarray(f0, np.float64)
    sp = np.asarray(sp, np.float64)
    ap = np.asarray(ap, np.float64)
    T = len(f0)
    N = (sp.shape[1] - 1) * 2
    y_length = int(T * frame_period * fs / 1000)
    fp = frame_period / 1000.0
    idx, ptime, shift, vuv = time_base(f0, fs, frame_period, y_length, N)
    P = len(idx)
    y = np.zeros(y_length)
    if P == 0:
        return dict(y=y, idx=idx, shift=shift, vuv=vuv, noise_size=np.zeros(0, np.int64), response=np.zeros((0, N)))
    nxt = idx[np.minimum(P - 1, np.arange(P) + 1)]
    noise_size = nxt - idx
    fl, ce, mix = _frame_mix(ptime, fp, T)
    mixc = np.where(fl == ce, 0.0, mix)[:, None]
    env = (1.0 - mixc) * np.abs(sp[fl]) + mixc * np.abs(sp[ce])                       # GetSpectralEnvelope
    safe = np.clip(ap, 0.001, 0.999999999999)
    ratio = ((1.0 - mixc) * safe[fl] + mixc * safe[ce]) ** 2                           # GetAperiodicRatio
    cur_vuv = vuv[idx]
    # periodic response
    voiced = (cur_vuv > 0.5) & ~(ratio[:, 0] > 0.999)
    mp = minimum_phase(np.log(env * (1.0 - ratio) + K_SAFE_MIN) / 2.0, N)
    k = np.arange(N // 2 + 1)
    coef = 2.0 * np.pi * shift[:, None] * fs / N
    re2 = np.cos(coef * k)
    im2 = np.sqrt(np.maximum(0.0, 1.0 - re2 * re2))
    mp = mp * (re2 - 1j * im2)
    per = np.fft.irfft(mp, n=N, axis=-1) * N                                            # unnormalised inverse (FFTW backward)
    per = np.concatenate([per[:, N // 2:], per[:, :N // 2]], axis=1)                     # fftshift
    dcr = dc_remover(N)
    dc = per[:, N // 2:].sum(axis=1, keepdims=True)
    per = np.concatenate([-dc * dcr[None, :N // 2], per[:, N // 2:] - dc * dcr[None, N // 2:]], axis=1)
    per[~voiced] = 0.0
    # aperiodic response: noise of noise_size samples (zero mean), coloured by the minimum-phase envelope
    table = randn_table(int(idx[-1] - idx[0]) + 1)
    noise = np.zeros((P, N))
    for i in range(P):
        n = int(noise_size[i])
        if n > 0:
            seg = table[idx[i] - idx[0]: idx[i] - idx[0] + n]
            noise[i, :min(n, N)] = (seg - seg.mean())[:N]
    nspec = np.fft.rfft(noise, axis=-1)
    col = np.where((cur_vuv != 0.0)[:, None], env * ratio, env)
    mpa = minimum_phase(np.log(col) / 2.0, N)
    aper = np.fft.irfft(mpa * nspec, n=N, axis=-1) * N
    aper = np.concatenate([aper[:, N // 2:], aper[:, :N // 2]], axis=1)
    resp = (per * np.sqrt(noise_size.astype(np.float64))[:, None] + aper) / N
    for i in range(P):                                                                    # overlap-add, pulse order
        off = int(idx[i]) - N // 2 + 1
        lo, hi = max(0, -off), min(N, y_length - off)
        if hi > lo:
            y[off + lo: off + hi] += resp[i, lo:hi]
    return dict(y=y, idx=idx, shift=shift, vuv=vuv, noise_size=noise_size, response=resp)


def synthesize(f0, sp, ap, fs, frame_period=5.0):
    """pyworld.synthesize(f0 [T], sp [T, N/2+1], ap [T, N/2+1], fs, frame_period ms) -> waveform [int(T * frame_period * fs / 1000)]."""
    return synthesize_parts(f0, sp, ap, fs, frame_period)["y"]
