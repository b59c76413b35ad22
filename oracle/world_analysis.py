"""WORLD analysis restatement (float64 numpy): DIO, CheapTrick, D4C, code_aperiodicity.  TEST INFRASTRUCTURE -- PARITY UNPINNED.

Reference call site: voice100/vocoder.py:61-87 --
    f0, time_axis = pyworld.dio(waveform, fs, f0_floor=80, f0_ceil=400, frame_period=10)
    spc = pyworld.cheaptrick(waveform, f0, time_axis, fs, fft_size=n_fft)
    ap = pyworld.d4c(waveform, f0, time_axis, fs, fft_size=n_fft)
    codeap = pyworld.code_aperiodicity(ap, fs)
The arithmetic lives in pyworld 0.3.2 (a Cython wrapper over M. Morise's C++ WORLD), which is neither in the reference tree nor in
this image; no pyworld output exists here.  This file restates the published algorithms --
  * DIO: M. Morise, H. Kawahara, H. Katayose, "Fast and reliable F0 estimation method based on the period extraction of vocal fold
    vibration of singing voice and speech", AES 35th Int. Conf., 2009 (band-wise low-pass filtering, four kinds of zero-crossing
    intervals per band, candidate = their mean, score = their spread, best band per frame, contour fixing);
  * CheapTrick: M. Morise, "CheapTrick, a spectral envelope estimator for high-quality speech synthesis", Speech Communication 67,
    2015 (F0-adaptive Hanning window of three periods, power spectrum, rectangular smoothing of width 2 F0 / 3, liftering with a sinc
    smoother and the q1 = -0.15 spectral recovery term);
  * D4C: M. Morise, "D4C, a band-aperiodicity estimator for high-quality speech synthesis", Speech Communication 84, 2016 (static group
    delay from two windowed centroids a quarter period apart, its band-limited power ratio per 3 kHz band; the "love train" voicing
    check on the 100 Hz - 4 kHz / 100 Hz - 7.9 kHz power ratio);
in the structure of the open-source implementation (dio.cpp, cheaptrick.cpp, d4c.cpp, codec.cpp) as its author documents it, and is
held to PROPERTIES (tests/test_oracle_world_analysis.py): F0 of synthetic harmonic signals, envelope recovery, aperiodicity of noise
against periodic signals, analysis -> synthesis round trips through oracle.world_synth.

The safeguard noise (`randn() * 1e-12` on every windowed sample, `|randn()| * eps` on every CheapTrick power bin) is part of the
algorithm -- it is what keeps digital silence finite -- and is drawn from WORLD's fixed sequence (oracle.world_synth.randn_table: the
generator is re-seeded at the start of cheaptrick() and of d4c()) in the order the C++ consumes it: frame by frame, sample by sample.
"""
import numpy as np

from .world_synth import interp1, number_of_aperiodicities, randn_table, K_FREQ_INTERVAL

K_CUTOFF = 50.0
K_MAXIMUM_VALUE = 100000.0
K_SAFE_MIN = 1e-12
K_DEFAULT_F0 = 500.0
K_FLOOR_F0_D4C = 47.0
K_EPS = 2.220446049250313e-16
K_LOG2 = 0.69314718055994529


def matlab_round(x):
    return int(x + 0.5) if x > 0 else int(x - 0.5)


def suitable_fft_size(sample):
    return int(2.0 ** (int(np.log(sample) / K_LOG2) + 1.0))


def nuttall_window(n):
    i = np.arange(n) / (n - 1.0)
    return 0.355768 - 0.487396 * np.cos(2 * np.pi * i) + 0.144232 * np.cos(4 * np.pi * i) - 0.012604 * np.cos(6 * np.pi * i)


# ---------------------------------------------------------------------------------------------------------------------------
# DIO
def samples_for_dio(fs, x_length, frame_period):
    return int(1000.0 * x_length / fs / frame_period) + 1


# DIGITAL SILENCE.  In a stretch of exact zeros WORLD's band signals are the rounding noise of its FFT convolution (relative 1e-16): noise-like,
# full of spurious zero crossings, so the F0 candidates there are garbage and get rejected -- an accident of the arithmetic that the
# algorithm nevertheless relies on: WITHOUT that noise a silent stretch has no events at all, interp1 EXTRAPOLATES the first real
# intervals across it, and a smooth, plausible, entirely fictitious F0 contour comes out (seen on the device, whose time-domain filters
# leave exact silence exactly silent: voiced frames deep inside zeroed lead-ins).  FFTW's rounding noise cannot be reproduced, so the
# noise floor is made explicit and deterministic: a dither of 1e-13 of the utterance's peak, from an integer hash of the sample index, is
# added to the input of the filters -- three orders above any implementation's rounding noise (silence behaves the same in the oracle and
# on the device, decision for decision), ten below anything an estimate could feel (F0 moves by ~1e-13 relative elsewhere).
DIO_DITHER = 1e-13


def dio_dither(n):
    """uniform in [-1, 1) from a 32-bit integer hash of the sample index (lowbias32), exactly reproducible anywhere"""
    h = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    m = np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(16))) * np.uint64(0x7FEB352D)) & m
    h = ((h ^ (h >> np.uint64(15))) * np.uint64(0x846CA68B)) & m
    h = h ^ (h >> np.uint64(16))
    return h.astype(np.float64) / 2147483648.0 - 1.0


def _low_cut_filter(n, fft_size):
    f = np.zeros(fft_size)
    i = np.arange(1, n + 1)
    f[:n] = 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (n + 1))
    f[:n] = -f[:n] / f[:n].sum()
    h = (n - 1) // 2
    f[fft_size - h:] = f[:h]
    f[:n] = f[h:h + n]
    f[0] += 1.0
    return f


def _zero_crossing_engine(sig, fs):
    n = len(sig)
    neg = np.nonzero((sig[:-1] > 0.0) & (sig[1:] <= 0.0))[0] + 1          # edges: index of the first non-positive sample
    if len(neg) < 2:
        return np.zeros(0), np.zeros(0)
    fine = neg - sig[neg - 1] / (sig[neg] - sig[neg - 1])
    intervals = fs / (fine[1:] - fine[:-1])
    locations = (fine[:-1] + fine[1:]) / 2.0 / fs
    return locations, intervals


def dio(x, fs, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, frame_period=5.0, speed=1, allowed_range=0.1, dither=None):
    """pyworld.dio(x, fs, f0_floor, f0_ceil, channels_in_octave, frame_period, speed, allowed_range) -> (f0 [T], temporal_positions [T]).
    speed = 1 only (no decimation: the reference does not pass it).
    dither: amplitude of the explicit noise floor relative to the utterance's peak (see DIGITAL SILENCE above).  None = DIO_DITHER,
    the value the DEVICE kernels use (csrc/world_analysis.hip), so oracle and device agree decision for decision in silence;
    dither = 0 is WORLD AS PUBLISHED -- no term that pyworld does not have -- and is what any comparison with pyworld vectors
    (tests/golden/thirdparty_world.npz, when it exists) must use."""
    if dither is None:
        dither = DIO_DITHER
    if speed != 1:
        raise NotImplementedError("decimation (speed > 1) is not restated")
    x = np.asarray(x, np.float64)
    x_length = len(x)
    nbands = 1 + int(np.log(f0_ceil / f0_floor) / K_LOG2 * channels_in_octave)
    boundary = f0_floor * 2.0 ** ((np.arange(nbands) + 1) / channels_in_octave)
    y_length = 1 + x_length
    fft_size = suitable_fft_size(y_length + matlab_round(fs / K_CUTOFF) * 2 + 1 + 4 * int(1.0 + fs / boundary[0] / 2.0))
    # spectrum for estimation: DC removed, low-cut filtered (50 Hz)
    y = np.zeros(fft_size)
    y[:x_length] = x
    y[:y_length] -= y[:y_length].sum() / y_length
    if dither:
        y[:y_length] += dio_dither(y_length) * (dither * np.abs(x).max() if x_length else 0.0)
    y_spec = np.fft.rfft(y)
    cutoff = matlab_round(fs / K_CUTOFF)
    y_spec = y_spec * np.fft.rfft(_low_cut_filter(cutoff * 2 + 1, fft_size))
    T = samples_for_dio(fs, x_length, frame_period)
    tpos = np.arange(T) * frame_period / 1000.0
    cand = np.zeros((nbands, T))
    score = np.full((nbands, T), K_MAXIMUM_VALUE)
    for b in range(nbands):
        half = matlab_round(fs / boundary[b] / 2.0)
        lp = np.zeros(fft_size)
        lp[:half * 4] = nuttall_window(half * 4)
        filt = np.fft.irfft(y_spec * np.fft.rfft(lp), n=fft_size)
        sig = filt[half * 2: half * 2 + y_length].copy()
        sets = []
        ok = True
        for kind in range(4):
            if kind == 0:
                s = sig
            elif kind == 1:
                s = -sig
            elif kind == 2:
                s = (-sig[:-1]) - (-sig[1:])            # differentiated AFTER the negation: negative-going zeros of it are the peaks
            else:
                s = -((-sig[:-1]) - (-sig[1:]))         # dips
            loc, itv = _zero_crossing_engine(s, fs)
            if len(itv) - 2 <= 0:                       # CheckEvent(count - 2): fewer than three intervals of any kind: no candidate in this band
                ok = False
                break
            sets.append(interp1(loc, itv, tpos))
        if not ok:
            continue
        sets = np.stack(sets)
        c = sets.mean(0)
        sc = np.sqrt(((sets - c) ** 2).sum(0) / 3.0)
        bad = (c > boundary[b]) | (c < boundary[b] / 2.0) | (c > f0_ceil) | (c < f0_floor)
        cand[b] = np.where(bad, 0.0, c)
        score[b] = np.where(bad, K_MAXIMUM_VALUE, sc)
    best = cand[np.argmin(score, axis=0), np.arange(T)]
    f0 = _fix_f0_contour(frame_period, nbands, cand, best, f0_floor, allowed_range)
    return f0, tpos


def _select_best_f0(cur, past, cand, target, allowed_range):
    ref = (cur * 3.0 - past) / 2.0
    err = np.abs(ref - cand[:, target])
    best = cand[int(np.argmin(err)), target]
    if abs(1.0 - best / ref) > allowed_range:
        return 0.0
    return best


def _fix_f0_contour(frame_period, nbands, cand, best, f0_floor, allowed_range):
    T = len(best)
    vrm = int(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1
    if T <= vrm:
        return np.zeros(T)
    # step 1: rapid jumps out
    base = best.copy()
    base[:vrm] = 0.0
    base[T - vrm:] = 0.0
    s1 = np.zeros(T)
    for i in range(vrm, T):
        s1[i] = base[i] if abs((base[i] - base[i - 1]) / (K_SAFE_MIN + base[i])) < allowed_range else 0.0
    # step 2: voiced sections shorter than the minimum out
    s2 = s1.copy()
    center = (vrm - 1) // 2
    for i in range(center, T - center):
        if np.any(s1[i - center: i + center + 1] == 0.0):
            s2[i] = 0.0
    pos, neg = [], []
    for i in range(1, T):
        if s2[i] == 0.0 and s2[i - 1] != 0.0:
            neg.append(i - 1)
        elif s2[i - 1] == 0.0 and s2[i] != 0.0:
            pos.append(i)
    # step 3: extend every section forward with the closest candidates
    s3 = s2.copy()
    for i, start in enumerate(neg):
        limit = T - 1 if i == len(neg) - 1 else neg[i + 1]
        for j in range(start, limit):
            s3[j + 1] = _select_best_f0(s3[j], s3[j - 1], cand, j + 1, allowed_range)
            if s3[j + 1] == 0.0:
                break
    # step 4: ... and backward
    s4 = s3.copy()
    for i in range(len(pos) - 1, -1, -1):
        limit = 1 if i == 0 else pos[i - 1]
        for j in range(pos[i], limit, -1):
            s4[j - 1] = _select_best_f0(s4[j], s4[j + 1], cand, j - 1, allowed_range)
            if s4[j - 1] == 0.0:
                break
    return s4


# ---------------------------------------------------------------------------------------------------------------------------
# shared pieces of CheapTrick and D4C
class _Randn:
    """WORLD's randn() stream from its re-seeded state, consumed in order."""

    def __init__(self, n):
        self.tab = np.asarray(randn_table(int(n)), np.float64)
        self.pos = 0

    def take(self, n):
        out = self.tab[self.pos: self.pos + n]
        assert len(out) == n, "randn stream exhausted"
        self.pos += n
        return out


def _d4c_window(x, fs, f0, position, window_type, ratio, rnd):
    half = matlab_round(ratio * fs / f0 / 2.0)
    base = np.arange(-half, half + 1)
    origin = matlab_round(position * fs + 0.001)
    safe = np.clip(origin + base, 0, len(x) - 1)
    pos = (2.0 * base / ratio) / fs
    if window_type == "hanning":
        w = 0.5 * np.cos(np.pi * pos * f0) + 0.5
    else:                                                                    # blackman
        w = 0.42 + 0.5 * np.cos(np.pi * pos * f0) + 0.08 * np.cos(np.pi * pos * f0 * 2.0)
    wav = x[safe] * w + rnd.take(len(base)) * K_SAFE_MIN
    wav = wav - w * (wav.sum() / w.sum())
    return wav, w


def _dc_correction(spec, f0, fs, fft_size):
    """Power below F0 is replaced by power + its mirror image about F0 (the window's main lobe folds the DC region)."""
    out = spec.copy()
    upper = 2 + int(f0 * fft_size / fs)
    axis = np.arange(upper) * fs / fft_size
    replica_n = upper - 1
    # interp1Q(f0 - axis[0], -fs / fft_size, input, upper + 1, axis, replica_n): input sampled on the grid f0 - k fs / fft_size
    xi = (axis[:replica_n] - (f0 - axis[0])) / (-fs / fft_size)
    base = np.minimum(xi.astype(np.int64), upper - 1)
    frac = xi - base
    inp = spec[:upper + 1]
    delta = np.diff(inp)
    rep = inp[base] + delta[np.minimum(base, len(delta) - 1)] * frac
    out[:replica_n] = spec[:replica_n] + rep
    return out


def _interp1q(x0, dx, y, xi):
    """interp1Q of WORLD: y sampled at x0 + k dx (dx may be negative), linear, queries inside the grid."""
    pos = (xi - x0) / dx
    base = pos.astype(np.int64)
    base = np.clip(base, 0, len(y) - 2)
    frac = pos - base
    return y[base] + (y[base + 1] - y[base]) * frac


def _linear_smoothing(spec, width, fs, fft_size):
    """Moving average of `width` Hz over the half spectrum (mirrored at both ends), by differences of the cumulative sum."""
    boundary = int(width * fft_size / fs) + 1
    half = fft_size // 2
    mirror = np.concatenate([spec[boundary:0:-1], spec[:half + 1], spec[half - 1: half - 1 - boundary: -1]])
    seg = np.cumsum(mirror * fs / fft_size)
    freq = np.arange(half + 1) / fft_size * fs
    origin = -(boundary - 0.5) * fs / fft_size
    low = _interp1q(origin, fs / fft_size, seg, freq - width / 2.0)
    high = _interp1q(origin, fs / fft_size, seg, freq + width / 2.0)
    return (high - low) / width


# ---------------------------------------------------------------------------------------------------------------------------
# CheapTrick
def cheaptrick_f0_floor(fs, fft_size):
    return 3.0 * fs / (fft_size - 3.0)


def cheaptrick(x, f0, temporal_positions, fs, q1=-0.15, f0_floor=71.0, fft_size=None):
    """pyworld.cheaptrick(x, f0, temporal_positions, fs, q1, f0_floor, fft_size) -> spectrogram [T, fft_size/2+1] (power).
    With fft_size given (the reference passes n_fft) the F0 floor becomes 3 fs / (fft_size - 3), as pyworld does."""
    x = np.asarray(x, np.float64)
    if fft_size is None:
        fft_size = int(2.0 ** (1.0 + int(np.log(3.0 * fs / f0_floor + 1) / K_LOG2)))
    else:
        f0_floor = cheaptrick_f0_floor(fs, fft_size)
    half = fft_size // 2
    out = np.zeros((len(f0), half + 1))
    quef = np.arange(half + 1) / float(fs)
    cf = np.where(np.asarray(f0) <= f0_floor, K_DEFAULT_F0, f0)
    rnd = _Randn(sum(2 * matlab_round(1.5 * fs / c) + 1 + half + 1 for c in cf))
    for i, (f, t) in enumerate(zip(f0, temporal_positions)):
        cf0 = K_DEFAULT_F0 if f <= f0_floor else f
        hw = matlab_round(1.5 * fs / cf0)
        base = np.arange(-hw, hw + 1)
        origin = matlab_round(t * fs + 0.001)
        safe = np.clip(origin + base, 0, len(x) - 1)
        pos = base / 1.5 / fs
        w = 0.5 * np.cos(np.pi * pos * cf0) + 0.5
        w = w / np.sqrt((w * w).sum())
        wav = x[safe] * w + rnd.take(len(base)) * K_SAFE_MIN
        wav = wav - w * (wav.sum() / w.sum())
        seg = np.zeros(fft_size)
        seg[:len(wav)] = wav
        power = np.abs(np.fft.rfft(seg)) ** 2
        power = _dc_correction(power, cf0, fs, fft_size)
        power = _linear_smoothing(power, cf0 * 2.0 / 3.0, fs, fft_size)
        power = power + np.abs(rnd.take(half + 1)) * K_EPS                 # AddInfinitesimalNoise
        # smoothing on the log axis + spectral recovery, both as lifters on the cepstrum
        smooth = np.ones(half + 1)
        smooth[1:] = np.sin(np.pi * cf0 * quef[1:]) / (np.pi * cf0 * quef[1:])
        comp = (1.0 - 2.0 * q1) + 2.0 * q1 * np.cos(2.0 * np.pi * quef * cf0)
        logp = np.log(power)
        cep = np.fft.rfft(np.concatenate([logp, logp[-2:0:-1]])).real
        out[i] = np.exp(np.fft.irfft(cep * smooth * comp, n=fft_size)[:half + 1])
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# D4C
def d4c(x, f0, temporal_positions, fs, threshold=0.85, fft_size=None):
    """pyworld.d4c(x, f0, temporal_positions, fs, threshold, fft_size) -> aperiodicity [T, fft_size/2+1] in (0, 1]."""
    x = np.asarray(x, np.float64)
    if fft_size is None:
        fft_size = int(2.0 ** (1.0 + int(np.log(3.0 * fs / 71.0 + 1) / K_LOG2)))
    T = len(f0)
    half = fft_size // 2
    ap = np.full((T, half + 1), 1.0 - K_SAFE_MIN)
    fft_d4c = int(2.0 ** (1.0 + int(np.log(4.0 * fs / K_FLOOR_F0_D4C + 1) / K_LOG2)))
    nb = number_of_aperiodicities(fs)
    wlen = int(K_FREQ_INTERVAL * fft_d4c / fs) * 2 + 1
    window = nuttall_window(wlen)
    f0 = np.asarray(f0, np.float64)
    voiced = f0[f0 != 0.0]
    draws = sum(2 * matlab_round(3.0 * fs / max(f, 40.0) / 2.0) + 1 for f in voiced) \
        + sum(3 * (2 * matlab_round(4.0 * fs / max(f, K_FLOOR_F0_D4C) / 2.0) + 1) for f in voiced)
    rnd = _Randn(draws)
    ap0 = _d4c_love_train(x, fs, f0, temporal_positions, rnd)
    coarse_f = np.concatenate([np.arange(nb + 1) * K_FREQ_INTERVAL, [fs / 2.0]])
    freq = np.arange(half + 1) * fs / fft_size
    for i in range(T):
        if f0[i] == 0 or ap0[i] <= threshold:
            continue
        cf0 = max(K_FLOOR_F0_D4C, f0[i])
        coarse = _d4c_general_body(x, fs, cf0, fft_d4c, temporal_positions[i], nb, window, wlen, rnd)
        coarse = np.minimum(0.0, coarse + (cf0 - 100.0) / 50.0)
        full = np.concatenate([[-60.0], coarse, [-K_SAFE_MIN]])
        ap[i] = 10.0 ** (interp1(coarse_f, full, freq) / 20.0)
    return ap


def _d4c_love_train(x, fs, f0, tpos, rnd):
    lowest = 40.0
    fft_size = int(2.0 ** (1.0 + int(np.log(3.0 * fs / lowest + 1) / K_LOG2)))
    b0 = int(np.ceil(100.0 * fft_size / fs))
    b1 = int(np.ceil(4000.0 * fft_size / fs))
    b2 = int(np.ceil(7900.0 * fft_size / fs))
    out = np.zeros(len(f0))
    for i in range(len(f0)):
        if f0[i] == 0.0:
            continue
        cf0 = max(f0[i], lowest)
        wav, _ = _d4c_window(x, fs, cf0, tpos[i], "blackman", 3.0, rnd)
        seg = np.zeros(fft_size)
        seg[:len(wav)] = wav
        p = np.abs(np.fft.rfft(seg)) ** 2
        p[:b0 + 1] = 0.0
        c = np.cumsum(p)
        out[i] = c[b1] / c[b2]
    return out


def _d4c_centroid(x, fs, f0, fft_size, position, rnd):
    wav, _ = _d4c_window(x, fs, f0, position, "blackman", 4.0, rnd)
    wav = wav / np.sqrt((wav * wav).sum())
    n = len(wav)
    seg = np.zeros(fft_size)
    seg[:n] = wav
    s1 = np.fft.rfft(seg)
    seg2 = np.zeros(fft_size)
    seg2[:n] = wav * (np.arange(n) + 1.0)
    s2 = np.fft.rfft(seg2)
    return s1.real * s2.real + s1.imag * s2.imag


def _d4c_general_body(x, fs, f0, fft_size, position, nb, window, wlen, rnd):
    half = fft_size // 2
    c1 = _d4c_centroid(x, fs, f0, fft_size, position - 0.25 / f0, rnd)
    c2 = _d4c_centroid(x, fs, f0, fft_size, position + 0.25 / f0, rnd)
    centroid = _dc_correction(c1 + c2, f0, fs, fft_size)
    wav, _ = _d4c_window(x, fs, f0, position, "hanning", 4.0, rnd)
    seg = np.zeros(fft_size)
    seg[:len(wav)] = wav
    power = np.abs(np.fft.rfft(seg)) ** 2
    power = _dc_correction(power, f0, fs, fft_size)
    power = _linear_smoothing(power, f0, fs, fft_size)
    gd = centroid / power
    gd = _linear_smoothing(gd, f0 / 2.0, fs, fft_size)
    gd = gd - _linear_smoothing(gd, f0, fs, fft_size)
    # band-wise: the group delay's fluctuation power in a 3 kHz Nuttall window, sorted, tail ratio
    boundary = matlab_round(fft_size * 8.0 / wlen)
    hw = wlen // 2
    coarse = np.zeros(nb)
    for i in range(nb):
        center = int(K_FREQ_INTERVAL * (i + 1) * fft_size / fs)
        seg = np.zeros(fft_size)
        seg[:wlen] = gd[center - hw: center - hw + wlen] * window
        p = np.sort(np.abs(np.fft.rfft(seg)) ** 2)
        c = np.cumsum(p)
        coarse[i] = 10.0 * np.log10(c[half - boundary - 1] / c[half])
    return coarse


# ---------------------------------------------------------------------------------------------------------------------------
def code_aperiodicity(ap, fs):
    """pyworld.code_aperiodicity(ap, fs) -> [T, nb] dB: the aperiodicity (dB) sampled at 3 kHz, 6 kHz, ..."""
    ap = np.asarray(ap, np.float64)
    nb = number_of_aperiodicities(fs)
    fft_size = (ap.shape[1] - 1) * 2
    freq = np.arange(ap.shape[1]) * fs / fft_size
    coarse_f = K_FREQ_INTERVAL * (np.arange(nb) + 1.0)
    return np.stack([interp1(freq, 20.0 * np.log10(a), coarse_f) for a in ap])
