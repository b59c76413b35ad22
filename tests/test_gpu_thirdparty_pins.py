"""The DEVICE kernels (csrc/world_analysis.hip, world.hip, mel.hip through voice100_amd.vocoder / voice100_amd.mel) against pyworld
0.3.2 / torchaudio 0.13.1 vectors, when tests/golden/thirdparty_*.npz exist (see tests/test_thirdparty_pins.py: skipped with the reason
until tests/golden/make_thirdparty_vectors.py has been run on a machine with the two wheels).  Bars: north_star's <= 1e-4 relative on
fp32 mel / WORLD features; voicing decisions equal outside exact digital silence (where the device adds its explicit noise floor)."""
import numpy as np
import pytest
import torch

from test_thirdparty_pins import _load, world_names, FS, FRAME_PERIOD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def voc():
    from voice100_amd.vocoder import WORLDVocoder
    return WORLDVocoder().cuda()


def test_device_analysis_matches_pyworld(voc):
    z = _load("thirdparty_world.npz")
    for n in world_names(z):
        x = torch.from_numpy(z[f"x_{n}"].astype(np.float32)).cuda()          # the reference hands float32 waveforms to encode()
        want = z[f"f0_{n}"]
        f0 = voc.dio(x, f0_floor=80.0, f0_ceil=400.0)[0].cpu().numpy()
        tpos = z[f"tpos_{n}"]
        xs = z[f"x_{n}"]
        silent = np.array([np.abs(xs[max(0, int(t * FS) - 400):int(t * FS) + 400]).max() == 0.0 for t in tpos])
        vu = (f0 > 0) != (want > 0)
        assert not (vu & ~silent).any(), (n, np.nonzero(vu & ~silent)[0][:10])
        both = (f0 > 0) & (want > 0)
        assert np.abs(f0[both] - want[both]).max(initial=0.0) <= 1e-4 * 400.0, n
        f0t = torch.from_numpy(want)[None].cuda()                             # pyworld's own contour: isolate the spectral stages
        sp = voc.cheaptrick(x, f0t)[0].cpu().numpy()
        assert np.abs(np.log(sp) - np.log(z[f"sp_{n}"])).max() < 1e-4, n
        ap, coded = voc.d4c(x, f0t)
        assert np.abs(ap[0].cpu().numpy() - z[f"ap_{n}"]).max() < 1e-4, n
        assert np.abs(coded[0].cpu().numpy() - z[f"codeap_{n}"]).max() < 1e-4 * np.abs(z[f"codeap_{n}"]).max(), n


def test_device_synthesis_matches_pyworld(voc):
    z = _load("thirdparty_world.npz")
    for n in world_names(z):
        f0 = torch.from_numpy(z[f"f0_{n}"].astype(np.float32))[None].cuda()
        sp = torch.from_numpy(z[f"sp_{n}"].astype(np.float32))[None].cuda()
        coded = torch.from_numpy(z[f"codeap_{n}"].astype(np.float32))[None].cuda()
        y, npulses = voc.synthesize(f0, sp, codeap=coded, f0_ceil=max(500.0, 2.0 * float(f0.max())) + 1.0)
        assert int(npulses[0]) > 0
        want = z[f"y_{n}"]
        got = y[0, :len(want)].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max(), n


def test_device_log_mel_matches_torchaudio():
    from voice100_amd.mel import MelSpectrogramAudioTransform
    z = _load("thirdparty_mel.npz")
    mel = MelSpectrogramAudioTransform().cuda()
    for n in ("1s", "10s"):
        got = mel.transform(torch.from_numpy(z[f"w_{n}"])[None].cuda())[0].cpu().numpy()
        want = z[f"logmel_{n}"]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max(), n
