"""Mel-cepstrum <-> log-spectrum matrices.  TEST INFRASTRUCTURE.

Reference: voice100/vocoder.py:105-141 (freqt, create_sp2mc_matrix,
create_mc2sp_matrix), float64 under the pinned numpy 1.24.1.
(Under numpy >= 2.0 `np.fft.irfft` keeps the float32 of `np.eye(..., float32)`,
so goldens generated in this container carry ~1e-7 of float32 rounding; this
restatement keeps the pinned float64 behaviour.)
"""
import numpy as np


def freqt(ceps: np.ndarray, order: int, alpha: float) -> np.ndarray:
    """All-pass frequency warp of each row of `ceps` to `order`+1 coefficients
    (SPTK freqt recursion, vocoder.py:126-141)."""
    ceps = np.asarray(ceps, dtype=np.float64)
    rows, m1 = ceps.shape
    c = np.zeros((rows, order + 1))
    beta = 1.0 - alpha * alpha
    for i in range(m1 - 1, -1, -1):        # feed coefficients last-to-first
        prev = c
        c = alpha * prev
        c[:, 0] += ceps[:, i]
        if order >= 1:
            c[:, 1] += beta * prev[:, 0]
        for j in range(2, order + 1):
            c[:, j] += prev[:, j - 1] - alpha * c[:, j - 1]
    return c


def sp2mc_matrix(fftlen: int, order: int, alpha: float) -> np.ndarray:
    """[fftlen/2+1, order+1]: logspc @ M = mel-cepstrum (vocoder.py:105-112)."""
    c = np.fft.irfft(np.eye(fftlen // 2 + 1, dtype=np.float64))
    c[:, 0] /= 2.0
    return freqt(c, order, alpha)


def mc2sp_matrix(fftlen: int, order: int, alpha: float) -> np.ndarray:
    """[order+1, fftlen/2+1]: mcep @ M = log-spectrum (vocoder.py:115-123)."""
    c = freqt(np.eye(order + 1, dtype=np.float64), fftlen // 2, -alpha)
    c[:, 0] *= 2.0
    c = np.concatenate([c, c[:, :0:-1]], axis=1)
    return np.fft.rfft(c).real


def vocoder_constants(sample_rate: int):
    """vocoder.py:28-41."""
    if sample_rate == 16000:
        return dict(mcep_dim=24, mcep_alpha=0.410, codeap_dim=1, n_fft=512)
    if sample_rate == 22050:
        return dict(mcep_dim=34, mcep_alpha=0.455, codeap_dim=2, n_fft=1024)
    raise ValueError("Unsupported sample rate")


def logspc_to_spc(logspc: np.ndarray, log_offset: float = 1e-15) -> np.ndarray:
    """vocoder.py:99."""
    return np.maximum(np.exp(logspc) - log_offset, 0)


def spc_to_logspc(spc: np.ndarray, log_offset: float = 1e-15) -> np.ndarray:
    """vocoder.py:71."""
    return np.log(spc + log_offset)
