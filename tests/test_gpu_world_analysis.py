"""WORLD analysis on the device (csrc/world_analysis.hip: DIO, CheapTrick, D4C, code_aperiodicity; vocoder.py:61-87) against the
float64 restatement in oracle/world_analysis.py.  PARITY UNPINNED (no pyworld in the image): the oracle states what is computed.

Bars: the device path is fp64 like the reference; discrete decisions (voiced / unvoiced, band choice) must agree frame for frame and
values to 1e-8 relative, far inside the 1e-4 of BASELINE.json's north_star.
"""
import numpy as np
import pytest
import torch

from oracle import world_analysis as wa
from oracle import world_synth as ws

pytestmark = pytest.mark.gpu
FS = 16000


def harmonic_signal(seconds=1.5, f_lo=150.0, swing=30.0, noise=1e-3, seed=1, lead=2400):
    t = np.arange(int(FS * seconds)) / FS
    f0 = f_lo + swing * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / FS
    x = sum(np.cos(k * ph) / k for k in range(1, 20)) * 0.1
    x[:lead] = 0
    x[-2400:] = 0
    x = x + np.random.default_rng(seed).standard_normal(len(x)) * noise
    return x.astype(np.float32)                     # the reference's waveforms are float32 tensors, converted to double


@pytest.fixture(scope="module")
def voc():
    from voice100_amd.vocoder import WORLDVocoder
    return WORLDVocoder().cuda()


@pytest.fixture(scope="module")
def case():
    x = harmonic_signal()
    xd = x.astype(np.float64)
    f0, tp = wa.dio(xd, FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
    sp = wa.cheaptrick(xd, f0, tp, FS, fft_size=512)
    ap = wa.d4c(xd, f0, tp, FS, fft_size=512)
    return x, f0, tp, sp, ap


def test_dio_matches_oracle(voc, case):
    x, f0, _, _, _ = case
    got = voc.dio(torch.from_numpy(x).cuda(), f0_floor=80.0, f0_ceil=400.0)[0].cpu().numpy()
    assert got.shape == f0.shape
    assert np.array_equal(got > 0, f0 > 0)
    assert (f0 > 0).sum() > 80
    assert np.abs(got - f0).max() < 1e-7


def test_dio_defaults_and_tones(voc):
    t = np.arange(FS) / FS
    for f in (90.0, 220.0, 380.0, 600.0):
        x = (np.sin(2 * np.pi * f * t) * 0.3 + np.random.default_rng(0).standard_normal(FS) * 1e-4).astype(np.float32)
        want, _ = wa.dio(x.astype(np.float64), FS, frame_period=10.0)              # pyworld defaults: 71 .. 800 Hz, 7 bands
        got = voc.dio(torch.from_numpy(x).cuda())[0].cpu().numpy()
        assert np.array_equal(got > 0, want > 0) and np.abs(got - want).max() < 1e-7, f
        assert np.abs(got[20:80] - f).max() < 0.5


def test_dio_ragged_batch_and_short_inputs(voc):
    xs = [harmonic_signal(1.5), harmonic_signal(0.9, 200.0, 20.0, seed=2), harmonic_signal(0.31, 120.0, 5.0, seed=3, lead=100)[:5000],
          np.zeros(300, np.float32)]
    L = max(len(x) for x in xs)
    batch = torch.zeros((len(xs), L))
    for i, x in enumerate(xs):
        batch[i, :len(x)] = torch.from_numpy(x)
    lengths = torch.tensor([len(x) for x in xs], dtype=torch.int32)
    got = voc.dio(batch.cuda(), lengths, f0_floor=80.0, f0_ceil=400.0).cpu().numpy()
    for i, x in enumerate(xs):
        want, _ = wa.dio(x.astype(np.float64), FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
        T = len(want)
        assert T == voc.frames(len(x))
        assert np.array_equal(got[i, :T] > 0, want > 0), i
        assert np.abs(got[i, :T] - want).max() < 1e-7, i
        assert not got[i, T:].any()


def test_cheaptrick_matches_oracle(voc, case):
    x, f0, _, sp, _ = case
    got = voc.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
    assert got.shape == sp.shape and np.isfinite(got).all() and (got > 0).all()
    d = np.abs(np.log(got) - np.log(sp))
    assert np.median(d) < 1e-8 and d.max() < 1e-6, (np.median(d), d.max())
    logsp = voc.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda(), log=True)[0].cpu().numpy()
    assert logsp.dtype == np.float32
    assert np.abs(logsp - np.log(sp + 1e-15).astype(np.float32)).max() < 2e-6


def test_cheaptrick_below_floor_and_silence(voc):
    x = harmonic_signal(0.5)
    tp = np.arange(51) * 0.01
    for f0 in (np.full(51, 90.0), np.zeros(51)):
        want = wa.cheaptrick(x.astype(np.float64), f0, tp, FS, fft_size=512)
        got = voc.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
        assert np.abs(np.log(got) - np.log(want)).max() < 1e-6
    # digital silence: what is left is WORLD's safeguard noise, drawn at the offsets the sequential code would draw it
    f0 = np.where((tp > 0.1) & (tp < 0.3), 160.0, 0.0)
    z = np.zeros(len(x), np.float32)
    want = wa.cheaptrick(z.astype(np.float64), f0, tp, FS, fft_size=512)
    got = voc.cheaptrick(torch.from_numpy(z).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
    assert np.isfinite(got).all() and got.max() < 1e-10
    assert np.median(np.abs(np.log(got) - np.log(want))) < 1e-3


def test_d4c_matches_oracle(voc, case):
    x, f0, _, _, ap = case
    got, coded = voc.d4c(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())
    got, coded = got[0].cpu().numpy(), coded[0].cpu().numpy()
    assert got.shape == ap.shape and np.isfinite(got).all()
    unv = np.isclose(ap[:, 0], 1.0 - 1e-12)
    assert np.array_equal(np.isclose(got[:, 0], 1.0 - 1e-12), unv)         # the same frames pass the love-train check
    assert (~unv).sum() > 60
    assert np.abs(got - ap).max() < 1e-8
    want_c = wa.code_aperiodicity(ap, FS)
    assert coded.shape == want_c.shape and np.abs(coded - want_c).max() < 1e-7
    _, c32 = voc.d4c(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda(), coded_only=True)
    assert c32.dtype == torch.float32 and np.abs(c32[0].cpu().numpy() - want_c).max() < 1e-5


def test_d4c_noise_is_aperiodic(voc):
    noise = (np.random.default_rng(0).standard_normal(FS) * 0.1).astype(np.float32)
    f0 = np.full(101, 150.0)
    ap, coded = voc.d4c(torch.from_numpy(noise).cuda(), torch.from_numpy(f0)[None].cuda())
    assert (ap > 0.999).all() and (coded.abs() < 1e-6).all()


def test_encode_matches_reference_recipe(voc):
    """WORLDVocoder.encode (vocoder.py:61-87): float32 CPU tensors, log(spc + 1e-15), coded aperiodicity."""
    x = harmonic_signal(1.2, 180.0, 25.0, seed=5)
    xd = x.astype(np.float64)
    f0, tp = wa.dio(xd, FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
    logspc = np.log(wa.cheaptrick(xd, f0, tp, FS, fft_size=512) + 1e-15)
    codeap = wa.code_aperiodicity(wa.d4c(xd, f0, tp, FS, fft_size=512), FS)
    g0, gl, gc = voc.encode(torch.from_numpy(x))
    assert not g0.is_cuda and g0.dtype == gl.dtype == gc.dtype == torch.float32
    assert g0.shape == (len(f0),) and gl.shape == logspc.shape and gc.shape == codeap.shape
    assert np.array_equal(g0.numpy(), f0.astype(np.float32))
    assert np.abs(gl.numpy() - logspc.astype(np.float32)).max() < 2e-6
    assert np.abs(gc.numpy() - codeap.astype(np.float32)).max() < 1e-5
    # mel-cepstral features: the reference multiplies in float64 and casts; the device GEMM is fp32
    from voice100_amd.vocoder import WORLDVocoder
    vm = WORLDVocoder(use_mcep=True).cuda()
    m0, mm, mc = vm.encode(torch.from_numpy(x))
    want = (logspc @ vm.sp2mc_matrix).astype(np.float32)
    assert mm.shape == want.shape and np.abs(mm.numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    assert torch.equal(m0, g0) and torch.equal(mc, gc)


def test_encode_decode_roundtrip(voc):
    """analysis -> synthesis on the device, re-analysed: F0 and envelope come back (the oracle's own round trip, on the GPU)."""
    x = harmonic_signal(1.5)
    f0, logspc, codeap = voc.encode(torch.from_numpy(x))
    y = voc.decode(f0, logspc, codeap)
    assert np.isfinite(y).all() and abs(len(y) - len(x)) <= 160
    f0b, logspcb, _ = voc.encode(torch.from_numpy(y.astype(np.float32)))
    n = min(len(f0), len(f0b))
    a, b = f0.numpy()[:n], f0b.numpy()[:n]
    both = (a > 0) & (b > 0)
    assert both.sum() > 0.9 * (a > 0).sum()
    assert np.median(np.abs(a[both] - b[both])) < 0.5
    tp = np.arange(n) * 0.01
    inner = both & (tp > 0.3) & (tp < tp[-1] - 0.3)
    d = (logspcb.numpy()[:n][inner][:, :200] - logspc.numpy()[:n][inner][:, :200]) * (10 / np.log(10))
    assert abs(d.mean()) < 1.0 and np.sqrt((d ** 2).mean()) < 3.0


def test_22k_shapes(voc):
    """22.05 kHz models (n_fft 1024, two aperiodicity bands): the same kernels at the other size the reference supports."""
    from voice100_amd.vocoder import WORLDVocoder
    v = WORLDVocoder(sample_rate=22050).cuda()
    fs = 22050
    t = np.arange(int(fs * 0.8)) / fs
    ph = 2 * np.pi * 170.0 * t
    x = (sum(np.cos(k * ph) / k for k in range(1, 30)) * 0.1 + np.random.default_rng(3).standard_normal(len(t)) * 1e-3).astype(np.float32)
    xd = x.astype(np.float64)
    f0, tp = wa.dio(xd, fs, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
    got = v.dio(torch.from_numpy(x).cuda(), f0_floor=80.0, f0_ceil=400.0)[0].cpu().numpy()
    assert np.array_equal(got > 0, f0 > 0) and np.abs(got - f0).max() < 1e-7
    sp = wa.cheaptrick(xd, f0, tp, fs, fft_size=1024)
    gs = v.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
    assert np.abs(np.log(gs) - np.log(sp)).max() < 1e-6
    ap = wa.d4c(xd, f0, tp, fs, fft_size=1024)
    ga, gc = v.d4c(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())
    assert np.abs(ga[0].cpu().numpy() - ap).max() < 1e-8
    assert gc.shape[-1] == 2 and np.abs(gc[0].cpu().numpy() - wa.code_aperiodicity(ap, fs)).max() < 1e-7


def test_random_speechlike_signals_agree_frame_for_frame(voc):
    """A small fuzz (tools/fuzz_world_analysis.py is the long one, profiles/r04_world_analysis_fuzz.txt): random harmonic + noise signals of
    random lengths in one ragged batch, one clipped, one with a third of DIGITAL silence in front -- where WORLD lives off the rounding noise of
    its FFT convolution and the restatement (oracle and device alike) off an explicit dither of 1e-13 of the peak: silence must come out
    unvoiced, not as the extrapolated contour a noise-free filter produces."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.bench_world_analysis import speechlike
    rng = np.random.default_rng(5)
    xs = []
    for i in range(6):
        x = speechlike(float(rng.uniform(0.4, 2.0)), FS, 300 + i)
        if i == 3:
            x = (x * 30).clip(-1, 1).astype(np.float32)
        if i == 4:
            x[: len(x) // 3] = 0                      # digital silence: defined by the explicit noise floor (oracle: DIGITAL SILENCE), see below
        xs.append(x)
    L = max(len(x) for x in xs)
    batch = torch.zeros((len(xs), L))
    for i, x in enumerate(xs):
        batch[i, :len(x)] = torch.from_numpy(x)
    lengths = torch.tensor([len(x) for x in xs], dtype=torch.int32)
    g0 = voc.dio(batch.cuda(), lengths, f0_floor=80.0, f0_ceil=400.0)
    gs = voc.cheaptrick(batch.cuda(), g0, lengths).cpu().numpy()
    ga, gc = voc.d4c(batch.cuda(), g0, lengths)
    g0, ga, gc = g0.cpu().numpy(), ga.cpu().numpy(), gc.cpu().numpy()
    voiced = 0
    for i, x in enumerate(xs):
        xd = x.astype(np.float64)
        f0, tp = wa.dio(xd, FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
        T = len(f0)
        voiced += int((f0 > 0).sum())
        assert np.array_equal(g0[i, :T] > 0, f0 > 0) and np.abs(g0[i, :T] - f0).max() < 1e-7, i
        sp = wa.cheaptrick(xd, f0, tp, FS, fft_size=512)
        assert np.abs(np.log(gs[i, :T]) - np.log(sp)).max() < 1e-6, i
        apo = wa.d4c(xd, f0, tp, FS, fft_size=512)
        assert np.abs(ga[i, :T] - apo).max() < 1e-8 and np.abs(gc[i, :T] - wa.code_aperiodicity(apo, FS)).max() < 1e-7, i
        assert not g0[i, T:].any() and not gs[i, T:].any() and not ga[i, T:].any() and not gc[i, T:].any()     # ragged rows end in zeros
        if i == 4:
            assert not g0[i, : len(x) // 3 // 160 - 2].any()            # nothing voiced inside the silence
    assert voiced > 200
    f0b, featb, capb = voc.encode_batch(batch.cuda(), lengths)
    for i, x in enumerate(xs):
        T = voc.frames(len(x))
        a, b_, c = voc.encode(torch.from_numpy(x))
        assert torch.equal(f0b[i, :T].cpu(), a) and torch.equal(featb[i, :T].cpu(), b_) and torch.equal(capb[i, :T].cpu(), c)
        assert not featb[i, T:].any() and not capb[i, T:].any()


def test_edge_cases_short_inputs_and_extreme_f0(voc):
    """One-frame inputs, inputs shorter than any analysis window, F0 contours at and beyond the estimators' floors and far above the
    search range (the per-frame estimators accept any contour): against the oracle."""
    rng = np.random.default_rng(9)
    for n in (1, 37, 159, 160, 700):
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        want, tp = wa.dio(x.astype(np.float64), FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
        got = voc.dio(torch.from_numpy(x).cuda(), f0_floor=80.0, f0_ceil=400.0)[0].cpu().numpy()
        assert got.shape == want.shape == (n * 100 // FS + 1,) and np.array_equal(got, want), n
        f0 = np.full(len(want), 123.0)
        sp = wa.cheaptrick(x.astype(np.float64), f0, tp, FS, fft_size=512)
        gs = voc.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
        assert np.abs(np.log(gs) - np.log(sp)).max() < 1e-6, n
        ap = wa.d4c(x.astype(np.float64), f0, tp, FS, fft_size=512)
        ga, gc = voc.d4c(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())
        assert np.abs(ga[0].cpu().numpy() - ap).max() < 1e-8, n
    x = harmonic_signal(0.6)
    tp = np.arange(61) * 0.01
    for f in (30.0, 40.0, 47.0, 60.0, 94.0, 95.0, 700.0, 1500.0):
        f0 = np.full(61, f)
        sp = wa.cheaptrick(x.astype(np.float64), f0, tp, FS, fft_size=512)
        gs = voc.cheaptrick(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())[0].cpu().numpy()
        assert np.isfinite(gs).all() and np.abs(np.log(gs) - np.log(sp)).max() < 1e-6, f
        ap = wa.d4c(x.astype(np.float64), f0, tp, FS, fft_size=512)
        ga, gc = voc.d4c(torch.from_numpy(x).cuda(), torch.from_numpy(f0)[None].cuda())
        ga = ga[0].cpu().numpy()
        assert np.isfinite(ga).all() and np.abs(ga - ap).max() < 1e-8, f
        assert np.abs(gc[0].cpu().numpy() - wa.code_aperiodicity(ap, FS)).max() < 1e-7, f


def test_device_on_the_references_own_world_synthesised_sample(voc):
    """The reference's docs/sample-en-1.wav (tests/golden/world_ref_samples.npz, README.md there: real pyworld-synthesised speech, int16): device == oracle frame for frame on
    it, and the device's own analysis -> synthesis keeps length and level."""
    import os
    x = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_ref_samples.npz"))["en1"].astype(np.float32) / 32768.0
    xd = x.astype(np.float64)
    f0, tp = wa.dio(xd, FS, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)
    g0 = voc.dio(torch.from_numpy(x).cuda(), f0_floor=80.0, f0_ceil=400.0)
    assert np.array_equal(g0[0].cpu().numpy() > 0, f0 > 0) and np.abs(g0[0].cpu().numpy() - f0).max() < 1e-7
    sp = wa.cheaptrick(xd, f0, tp, FS, fft_size=512)
    gs = voc.cheaptrick(torch.from_numpy(x).cuda(), g0)[0].cpu().numpy()
    assert np.abs(np.log(gs) - np.log(sp)).max() < 1e-6
    ap = wa.d4c(xd, f0, tp, FS, fft_size=512)
    ga, gc = voc.d4c(torch.from_numpy(x).cuda(), g0)
    assert np.abs(ga[0].cpu().numpy() - ap).max() < 1e-8
    a, b, c = voc.encode(torch.from_numpy(x))
    y = voc.decode(a[:718], b[:718], c[:718])
    assert len(y) == len(x) and 0.8 < np.abs(y).max() / np.abs(x).max() < 1.25


@pytest.mark.parametrize("name,windows", [("ja1_head", 3), ("en2_head", 1)])
def test_device_reproduces_the_reference_samples_unvoiced_lead_in(voc, name, windows):
    """tests/test_oracle_world_analysis.py::test_unvoiced_lead_in_... on the DEVICE: encode() -> decode() of the reference's own file gives back
    its unvoiced lead-in sample by sample (WORLD's randn bursts through the minimum-phase envelope), at the reference's level."""
    import os
    x = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_ref_samples.npz"))[name].astype(np.float32) / 32768.0
    f0, feat, codeap = voc.encode(torch.from_numpy(x))
    T = len(x) // 160
    y = voc.decode(f0[:T], feat[:T], codeap[:T])
    assert len(y) == T * 160
    for i in range(windows):
        c = np.corrcoef(x[320 * i: 320 * (i + 1)], y[320 * i: 320 * (i + 1)])[0, 1]
        assert c > 0.85, (i, c)
    assert 0.4 < np.sqrt((y[:320] ** 2).mean()) / np.sqrt((x[:320].astype(np.float64) ** 2).mean()) < 2.5
