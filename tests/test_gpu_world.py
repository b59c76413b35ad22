"""WORLD synthesis on the device (csrc/world.hip; vocoder.py:100-101) against the float64 restatement oracle/world_synth.py.
PARITY UNPINNED (pyworld's C++ is not in the reference tree): the oracle itself is held to properties in tests/test_oracle_world.py;
here the HIP path must reproduce it -- the same number of pulses at the same samples (a pulse one sample off would show as an O(1)
error), waveform within 1e-4 of the utterance's peak (north_star's fp32 feature bar) -- and show the same properties on its own
output.  Everything goes through the C ABI (v100_world_*)."""
import numpy as np
import pytest
import torch

from oracle import world_synth as W

pytestmark = pytest.mark.gpu
FS, N = 16000, 512
TOL = 1e-4


def _formant_sp(T, level=1e-2, rng=None):
    k = np.arange(N // 2 + 1)
    sp = np.tile(level * (1 + 4 * np.exp(-((k * FS / N - 1500) / 300.0) ** 2)), (T, 1))
    if rng is not None:
        sp = sp * np.exp(0.3 * rng.randn(T, 1)) * (1 + 0.2 * rng.rand(T, N // 2 + 1))
    return sp


def _dev(a, cuda, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(cuda)


def _cases():
    rng = np.random.RandomState(7)
    T = 120
    f_mixed = np.where((np.arange(T) // 20) % 2 == 0, 0.0, 120.0 + 60.0 * np.sin(np.arange(T) / 9.0))
    return {
        "voiced_210": (np.full(T, 210.0), np.full((T, 1), -30.0), _formant_sp(T)),
        "unvoiced": (np.zeros(T), np.zeros((T, 1)), _formant_sp(T, 0.04)),
        "mixed": (f_mixed, np.where(f_mixed[:, None] > 0, -25.0 + 10 * rng.rand(T, 1), -0.1 * rng.rand(T, 1)), _formant_sp(T, rng=rng)),
        "low_f0_and_gate": (np.where(np.arange(T) < 60, 33.0, 20.0), np.full((T, 1), -12.0), _formant_sp(T, rng=rng)),   # 20 Hz < fs/512 + 1: unvoiced
        "two_frames": (np.array([150.0, 160.0]), np.full((2, 1), -20.0), _formant_sp(2)),
    }


def test_randn_table_is_worlds_sequence(cuda):
    from voice100_amd.vocoder import WORLDVocoder
    t = WORLDVocoder._randn_table(5000, cuda).cpu().numpy()
    assert np.array_equal(t[:5000], W.randn_table(5000).astype(np.float32))


def test_decode_aperiodicity_vs_oracle(cuda):
    from voice100_amd.vocoder import WORLDVocoder
    v = WORLDVocoder()
    rng = np.random.RandomState(1)
    cod = np.concatenate([-40 * rng.rand(50, 1), np.array([[0.0], [-0.4], [-0.5], [-0.6], [-60.0]])])
    got = v.decode_aperiodicity(_dev(cod, cuda)).cpu().numpy()
    ref = W.decode_aperiodicity(cod.astype(np.float32).astype(np.float64), FS, N)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-6
    got3 = v.decode_aperiodicity(_dev(cod.reshape(5, 11, 1), cuda)).cpu().numpy()
    assert np.array_equal(got3.reshape(-1, 257), got)


@pytest.mark.parametrize("name", list(_cases()))
def test_synthesize_vs_oracle(cuda, name):
    from voice100_amd.vocoder import WORLDVocoder
    f0, cod, sp = _cases()[name]
    f0 = f0.astype(np.float32).astype(np.float64)
    sp = sp.astype(np.float32).astype(np.float64)
    v = WORLDVocoder()
    ap_d = v.decode_aperiodicity(_dev(cod, cuda))
    ref = W.synthesize_parts(f0, sp, ap_d.cpu().numpy().astype(np.float64), FS, 10.0)
    y, n = v.synthesize(_dev(f0[None], cuda), _dev(sp[None], cuda), ap_d[None].contiguous())
    assert int(n[0]) == len(ref["idx"])                       # same pulse count: the time base is reproduced sample for sample
    got = y[0].double().cpu().numpy()
    assert got.shape == ref["y"].shape
    peak = np.abs(ref["y"]).max()
    assert np.isfinite(got).all() and np.abs(got - ref["y"]).max() <= TOL * peak, (name, np.abs(got - ref["y"]).max() / peak)


def test_synthesize_batch_ragged_frames_equals_single(cuda):
    """A batch with frames [B]: each row equals the utterance synthesised alone (its own length, zero beyond), bit for bit."""
    from voice100_amd.vocoder import WORLDVocoder
    rng = np.random.RandomState(11)
    v = WORLDVocoder()
    T = 90
    lens = [90, 37, 2, 64]
    f0 = np.where(rng.rand(4, T) < 0.3, 0.0, 100 + 150 * rng.rand(4, T)).astype(np.float32)
    sp = np.stack([_formant_sp(T, rng=rng) for _ in lens]).astype(np.float32)
    cod = np.where(f0[..., None] > 0, -30 * rng.rand(4, T, 1), 0.0).astype(np.float32)
    ap = v.decode_aperiodicity(_dev(cod, cuda))
    y, n = v.synthesize(_dev(f0, cuda), _dev(sp, cuda), ap, frames=torch.tensor(lens, dtype=torch.int32))
    assert y.shape == (4, int(T * 10.0 * FS / 1000))
    for b, L in enumerate(lens):
        yl = int(L * 10.0 * FS / 1000)
        y1, n1 = v.synthesize(_dev(f0[b:b + 1, :L], cuda), _dev(sp[b:b + 1, :L], cuda), ap[b:b + 1, :L].clone())
        assert int(n1[0]) == int(n[b])
        assert torch.equal(y[b, :yl], y1[0]) and float(y[b, yl:].abs().max() if yl < y.shape[1] else 0.0) == 0.0
        ref = W.synthesize(f0[b, :L].astype(np.float64), sp[b, :L].astype(np.float64), ap[b, :L].cpu().numpy().astype(np.float64), FS, 10.0)
        assert np.abs(y1[0].double().cpu().numpy() - ref).max() <= TOL * np.abs(ref).max()


def test_synthesize_pulse_overflow_is_loud_and_config_size(cuda):
    from voice100_amd.vocoder import WORLDVocoder
    v = WORLDVocoder()
    T = 50
    f0 = torch.full((1, T), 3000.0, device=cuda)            # 3 kHz "pitch": far more pulses than f0_ceil = 1000 makes room for
    sp = _dev(_formant_sp(T)[None], cuda)
    ap = v.decode_aperiodicity(torch.full((1, T, 1), -20.0, device=cuda))
    y, n = v.synthesize(f0, sp, ap)
    assert int(n[0]) == -1 and torch.isnan(y).all()
    y, n = v.synthesize(f0, sp, ap, f0_ceil=3200.0)
    assert int(n[0]) > 0 and torch.isfinite(y).all()
    # configs[2] size: B = 16 x 1023 frames (10.2 s each) in one call, against the oracle on two of the rows
    rng = np.random.RandomState(5)
    B, T = 16, 1023
    f0 = np.where(np.sin(np.arange(T)[None] / 40.0 + rng.rand(B, 1) * 6) > 0.2, 0.0, 90 + 160 * rng.rand(B, 1) + 20 * np.sin(np.arange(T)[None] / 7.0)).astype(np.float32)
    sp = (_formant_sp(T)[None] * np.exp(0.3 * rng.randn(B, T, 1))).astype(np.float32)
    cod = np.where(f0[..., None] > 0, -10 - 25 * rng.rand(B, T, 1), 0.0).astype(np.float32)
    ap = v.decode_aperiodicity(_dev(cod, cuda))
    y, n = v.synthesize(_dev(f0, cuda), _dev(sp, cuda), ap)
    assert y.shape == (B, 163680) and (n > 0).all() and torch.isfinite(y).all()
    for b in (0, 9):
        ref = W.synthesize_parts(f0[b].astype(np.float64), sp[b].astype(np.float64), ap[b].cpu().numpy().astype(np.float64), FS, 10.0)
        assert int(n[b]) == len(ref["idx"])
        assert np.abs(y[b].double().cpu().numpy() - ref["y"]).max() <= TOL * np.abs(ref["y"]).max()


def test_device_output_has_the_pitch_the_envelope_and_the_noise_level(cuda):
    """The property tests of tests/test_oracle_world.py on the DEVICE output (in lieu of pyworld vectors)."""
    from voice100_amd.vocoder import WORLDVocoder
    v = WORLDVocoder()
    T, F0 = 200, 210.0
    sp = _formant_sp(T)
    ap = v.decode_aperiodicity(torch.full((1, T, 1), -30.0, device=cuda))
    y, _ = v.synthesize(torch.full((1, T), F0, device=cuda), _dev(sp[None], cuda), ap)
    y = y[0].double().cpu().numpy()[8000:24000]
    ac = np.correlate(y, y, "full")[len(y) - 1:]
    assert int(np.argmax(ac[40:200])) + 40 == 76
    S = np.abs(np.fft.rfft(y * np.hanning(len(y))))
    harm = np.array([S[int(round(h * F0 * len(y) / FS))] for h in range(1, 30)])
    want = np.sqrt(np.interp(np.arange(1, 30) * F0, np.arange(N // 2 + 1) * FS / N, sp[0]))
    rel = (harm / harm[2]) / (want / want[2])
    assert np.all(np.abs(rel[:20] - 1.0) < 0.12)
    apu = v.decode_aperiodicity(torch.zeros((1, T, 1), device=cuda))
    yu, _ = v.synthesize(torch.zeros((1, T), device=cuda), torch.full((1, T, 257), 0.04, device=cuda), apu)
    yu = yu[0].double().cpu().numpy()[2000:-2000]
    assert abs(yu.var() / 0.04 - 1.0) < 0.1 and abs(float(np.dot(yu[1:], yu[:-1]) / np.dot(yu, yu))) < 0.1


def test_decode_is_a_drop_in_and_the_pipeline_ends_in_a_waveform(cuda):
    """WORLDVocoder.decode(f0, features, codeap) -> float64 numpy waveform (vocoder.py:89-102), CPU or CUDA tensors in; the configs[2]
    chain (TTSPipeline) returns a finite waveform of 160 samples per WORLD frame."""
    from voice100_amd.vocoder import WORLDVocoder
    from voice100_amd.tts import TextToAlignTextModel, AlignTextToAudioModel
    from voice100_amd.infer import TTSPipeline
    rng = np.random.RandomState(2)
    T = 80
    for use_mcep in (False, True):
        v = WORLDVocoder(use_mcep=use_mcep).to(cuda)
        f0 = torch.from_numpy(np.where(rng.rand(T) < 0.3, 0.0, 120 + 80 * rng.rand(T)).astype(np.float32))
        feat = torch.from_numpy((0.3 * rng.randn(T, 25)).astype(np.float32)) if use_mcep else torch.from_numpy(np.log(_formant_sp(T) + 1e-15).astype(np.float32))
        cod = torch.from_numpy(np.where(f0.numpy()[:, None] > 0, -20.0, 0.0).astype(np.float32))
        w = v.decode(f0, feat, cod)                                    # CPU tensors, as the reference's scripts pass them
        assert isinstance(w, np.ndarray) and w.dtype == np.float64 and w.shape == (int(T * 10.0 * FS / 1000),) and np.isfinite(w).all()
        w2 = v.decode(f0.to(cuda), feat.to(cuda), cod.to(cuda))
        assert np.array_equal(w, w2)
        logspc = feat.double().numpy() @ v.mc2sp_matrix if use_mcep else feat.double().numpy()
        spc = np.maximum(np.exp(logspc) - 1e-15, 0)
        ref = W.synthesize(f0.double().numpy(), spc, W.decode_aperiodicity(cod.double().numpy(), FS, N), FS, 10.0)
        assert np.abs(w - ref).max() <= 5e-4 * np.abs(ref).max()       # fp32 spectrum (GEMM + exp on the device) in front of the synthesis
    torch.manual_seed(4)
    al = TextToAlignTextModel(vocab_size=29, hidden_size=64).to(cuda).eval()
    with torch.no_grad():
        al.layers[4].bias.copy_(torch.tensor([0.6931, 1.3863], device=cuda))
    au = AlignTextToAudioModel(vocab_size=29, hidden_size=64, use_mcep=True).to(cuda).eval()
    with torch.no_grad():                                   # feature statistics of a real voice, so that voiced pulses occur
        au.norm.f0_mean.fill_(150.0); au.norm.f0_std.fill_(30.0); au.norm.codeap_mean.fill_(-12.0); au.norm.codeap_std.fill_(3.0)
    chain = TTSPipeline(al, au, WORLDVocoder(use_mcep=True).to(cuda))
    text = torch.randint(1, 29, (3, 20), device=cuda)
    out = chain(text, torch.tensor([20, 11, 16], device=cuda))
    assert out["wave"].shape[0] == 3 and out["wave"].shape[1] == int(out["f0"].shape[1] * 10.0 * FS / 1000)
    assert torch.isfinite(out["wave"]).all() and (out["n_pulses"] > 0).all()
    assert torch.equal(out["wave_len"].cpu(), (out["frames"].cpu() * 160).to(torch.int64))
    for b in range(3):
        assert float(out["wave"][b, int(out["wave_len"][b]):].abs().sum()) == 0.0


def test_synthesize_from_coded_aperiodicity_vs_double_oracle(cuda):
    """codeap= : the band aperiodicity decoded inside the kernel (as 1 - a, nothing near 1 rounded to fp32) against the oracle fed the
    DOUBLE-precision decode -- the reference's own order of operations (vocoder.py:100-101 runs both steps in double)."""
    from voice100_amd.vocoder import WORLDVocoder
    rng = np.random.RandomState(21)
    v = WORLDVocoder()
    B, T = 3, 300
    f0 = np.where(np.sin(np.arange(T)[None] / 25.0 + rng.rand(B, 1) * 6) > 0.3, 0.0, 100 + 120 * rng.rand(B, 1) + 15 * np.sin(np.arange(T)[None] / 5.0)).astype(np.float32)
    sp = np.stack([_formant_sp(T, rng=rng) for _ in range(B)]).astype(np.float32)
    cod = np.where(f0[..., None] > 0, -3 - 30 * rng.rand(B, T, 1), -0.3 * rng.rand(B, T, 1)).astype(np.float32)
    y, n = v.synthesize(_dev(f0, cuda), _dev(sp, cuda), codeap=_dev(cod, cuda))
    for b in range(B):
        ap64 = W.decode_aperiodicity(cod[b].astype(np.float64), FS, N)
        ref = W.synthesize_parts(f0[b].astype(np.float64), sp[b].astype(np.float64), ap64, FS, 10.0)
        assert int(n[b]) == len(ref["idx"])
        err = np.abs(y[b].double().cpu().numpy() - ref["y"]).max() / np.abs(ref["y"]).max()
        assert err <= TOL, (b, err)
    with pytest.raises(ValueError):
        v.synthesize(_dev(f0, cuda), _dev(sp, cuda))


def test_synthesize_22k_fft1024_vs_oracle(cuda):
    """The other sample rate WORLDVocoder supports (vocoder.py:34-41: 22.05 kHz, n_fft 1024, two aperiodicity bands): the general-size
    fp64 pulse kernel against the oracle, from a decoded aperiodicity tensor and from the coded bands."""
    from voice100_amd.vocoder import WORLDVocoder
    fs, n = 22050, 1024
    rng = np.random.RandomState(11)
    v = WORLDVocoder(sample_rate=fs)
    assert v.n_fft == n and v.codeap_dim == 2
    B, T = 2, 150
    f0 = np.where(np.sin(np.arange(T)[None] / 20.0 + rng.rand(B, 1) * 6) > 0.3, 0.0, 110 + 100 * rng.rand(B, 1) + 12 * np.sin(np.arange(T)[None] / 5.0)).astype(np.float32)
    k = np.arange(n // 2 + 1)
    sp = (1e-2 * (1 + 4 * np.exp(-((k * fs / n - 1500) / 300.0) ** 2))[None, None] * np.exp(0.3 * rng.randn(B, T, 1)) * (1 + 0.2 * rng.rand(B, T, n // 2 + 1))).astype(np.float32)
    cod = np.where(f0[..., None] > 0, -3 - 30 * rng.rand(B, T, 2), -0.3 * rng.rand(B, T, 2)).astype(np.float32)
    frames = torch.tensor([T, T - 37], dtype=torch.int32)
    y, npl = v.synthesize(_dev(f0, cuda), _dev(sp, cuda), codeap=_dev(cod, cuda), frames=frames)
    assert y.shape == (B, int(T * 10.0 * fs / 1000))
    for b in range(B):
        t = int(frames[b])
        ap64 = W.decode_aperiodicity(cod[b, :t].astype(np.float64), fs, n)
        ref = W.synthesize_parts(f0[b, :t].astype(np.float64), sp[b, :t].astype(np.float64), ap64, fs, 10.0)
        assert int(npl[b]) == len(ref["idx"])
        got = y[b].double().cpu().numpy()
        err = np.abs(got[:len(ref["y"])] - ref["y"]).max() / np.abs(ref["y"]).max()
        assert err <= 1e-6, (b, err)                            # fp64 kernel, fp32 only in the stored responses
        assert not got[len(ref["y"]):].any()
    ap32 = v.decode_aperiodicity(_dev(cod, cuda))
    assert ap32.shape == (B, T, n // 2 + 1)
    y2, n2 = v.synthesize(_dev(f0, cuda), _dev(sp, cuda), ap32)
    ref = W.synthesize_parts(f0[0].astype(np.float64), sp[0].astype(np.float64), ap32[0].double().cpu().numpy(), fs, 10.0)
    assert int(n2[0]) == len(ref["idx"])
    assert np.abs(y2[0].double().cpu().numpy() - ref["y"]).max() / np.abs(ref["y"]).max() <= 1e-6
    # decode() as the reference's drop-in at this rate
    w = v.decode(torch.from_numpy(f0[0]), torch.log(torch.from_numpy(sp[0]) + 1e-15), torch.from_numpy(cod[0]))
    assert w.dtype == np.float64 and w.shape == (int(T * 10.0 * fs / 1000),) and np.isfinite(w).all()


@pytest.mark.parametrize("seed", [3, 4])
def test_synthesize_long_utterance_many_chunks(cuda, seed):
    """25 s (40 chunks of the chained time base: the running phase handed from workgroup to workgroup, binade crossings up to 2^16, ties):
    the pulse count and the waveform must still match the sequential float64 oracle."""
    from voice100_amd.vocoder import WORLDVocoder
    rng = np.random.RandomState(seed)
    v = WORLDVocoder()
    T = 2500
    f0 = np.where(np.sin(np.arange(T) / 37.0 + rng.rand() * 6) > 0.3, 0.0, 80 + 250 * rng.rand() + 25 * np.sin(np.arange(T) / 6.0)).astype(np.float32)
    if seed == 4:
        f0[:] = np.where(f0 > 0, 200.0, 0.0)                 # constant voiced F0 / the unvoiced default: constant increments (tie stretches)
    sp = _formant_sp(T, rng=rng).astype(np.float32)
    cod = np.where(f0[:, None] > 0, -5 - 30 * rng.rand(T, 1), -0.3 * rng.rand(T, 1)).astype(np.float32)
    y, n = v.synthesize(_dev(f0[None], cuda), _dev(sp[None], cuda), codeap=_dev(cod[None], cuda))
    ap64 = W.decode_aperiodicity(cod.astype(np.float64), FS, N)
    ref = W.synthesize_parts(f0.astype(np.float64), sp.astype(np.float64), ap64, FS, 10.0)
    assert int(n[0]) == len(ref["idx"])
    err = np.abs(y[0].double().cpu().numpy() - ref["y"]).max() / np.abs(ref["y"]).max()
    assert err <= TOL, err
