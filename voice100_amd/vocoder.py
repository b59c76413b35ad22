"""WORLD vocoder glue (voice100/vocoder.py:14-141): the reference-owned arithmetic -- mel-cepstrum <-> log
spectrum matrices (host, float64, built once), their application and the log/exp/clip steps -- with the
per-frame products on the GPU.  DIO / CheapTrick / D4C / synthesis live in pyworld (C++ WORLD), which is not
part of this hot path: encode()/decode() call it when installed and raise otherwise.
"""
from typing import Tuple

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import functional as F_

__all__ = ["WORLDVocoder", "create_sp2mc_matrix", "create_mc2sp_matrix", "freqt"]


def freqt(ceps: np.ndarray, order: int = 25, alpha: float = 0.0) -> np.ndarray:
    """SPTK-compatible frequency warp of the rows of `ceps` (vocoder.py:126-141), vectorised over rows."""
    ceps = np.asarray(ceps, dtype=np.float64)
    out = np.zeros((ceps.shape[0], order + 1))
    b = 1.0 - alpha * alpha
    for x in ceps.T[::-1]:                       # last coefficient first
        prev = out
        out = alpha * prev
        out[:, 0] += x
        if order >= 1:
            out[:, 1] += b * prev[:, 0]
        for j in range(2, order + 1):
            out[:, j] += prev[:, j - 1] - alpha * out[:, j - 1]
    return out


def create_sp2mc_matrix(fftlen: int, order: int, alpha: float) -> np.ndarray:
    """[fftlen/2+1, order+1] with logspc @ M = mcep (vocoder.py:105-112)."""
    c = np.fft.irfft(np.eye(fftlen // 2 + 1, dtype=np.float64))
    c[:, 0] /= 2.0
    return freqt(c, order, alpha)


def create_mc2sp_matrix(fftlen: int, order: int, alpha: float) -> np.ndarray:
    """[order+1, fftlen/2+1] with mcep @ M = logspc (vocoder.py:115-123)."""
    c = freqt(np.eye(order + 1, dtype=np.float64), fftlen // 2, -alpha)
    c[:, 0] *= 2.0
    c = np.concatenate([c, c[:, :0:-1]], axis=1)
    return np.fft.rfft(c).real


class WORLDVocoder(nn.Module):
    def __init__(self, sample_rate: int = 16000, frame_period: float = 10.0, n_fft: int = None, use_mcep: bool = False,
                 log_offset: float = 1e-15) -> None:
        super().__init__()
        self.sample_rate = sample_rate
        self.frame_period = frame_period
        self.n_fft = n_fft
        if sample_rate == 16000:
            self.mcep_dim, self.mcep_alpha, self.codeap_dim = 24, 0.410, 1
            if self.n_fft is None:
                self.n_fft = 512
        elif sample_rate == 22050:
            self.mcep_dim, self.mcep_alpha, self.codeap_dim = 34, 0.455, 2
            if self.n_fft is None:
                self.n_fft = 1024
        else:
            raise ValueError("Unsupported sample rate")
        self.use_mcep = use_mcep
        if use_mcep:
            self.sp2mc_matrix = create_sp2mc_matrix(self.n_fft, self.mcep_dim, alpha=self.mcep_alpha)
            self.mc2sp_matrix = create_mc2sp_matrix(self.n_fft, self.mcep_dim, alpha=self.mcep_alpha)
            # transposed fp32 copies: the per-frame products run as [out x in] x [in x T] GEMMs on the GPU
            self.register_buffer("_sp2mc_t", torch.from_numpy(np.ascontiguousarray(self.sp2mc_matrix.T).astype(np.float32)), persistent=False)
            self.register_buffer("_mc2sp_t", torch.from_numpy(np.ascontiguousarray(self.mc2sp_matrix.T).astype(np.float32)), persistent=False)
        else:
            self.sp2mc_matrix = None
            self.mc2sp_matrix = None
        self.log_offset = log_offset

    @property
    def output_dims(self) -> Tuple[int, int, int]:
        return (1, self.mcep_dim + 1, self.codeap_dim) if self.use_mcep else (1, self.n_fft // 2 + 1, self.codeap_dim)

    # ---- reference-owned glue on the GPU ----------------------------------------------------------
    def _frames_matmul(self, x_tf: torch.Tensor, mat_t: torch.Tensor) -> torch.Tensor:
        """x [T, F] @ M [F, G] -> [T, G], as GEMM(M^T [G x F], x^T [F x T]) on the exact-fp32 MFMA kernel."""
        F_._check(x_tf, "WORLDVocoder")
        T, Fdim = x_tf.shape
        G = mat_t.shape[0]
        xt = F_.transpose_last2(x_tf[None].contiguous())                 # [1, F, T]
        y = torch.empty((1, G, T), dtype=torch.float32, device=x_tf.device)
        F_._pw_gemm(mat_t, None, xt, y, G, Fdim, T, 1, False)
        return F_.transpose_last2(y)[0]                                   # [T, G]

    @torch.no_grad()
    def logspc_to_mcep(self, logspc: torch.Tensor) -> torch.Tensor:
        """vocoder.py:76: mcep = logspc @ sp2mc."""
        return self._frames_matmul(logspc, self._sp2mc_t)

    @torch.no_grad()
    def mcep_to_logspc(self, mcep: torch.Tensor) -> torch.Tensor:
        """vocoder.py:95: logspc = mcep @ mc2sp."""
        return self._frames_matmul(mcep, self._mc2sp_t)

    @torch.no_grad()
    def logspc_to_spc(self, logspc: torch.Tensor) -> torch.Tensor:
        """vocoder.py:99: spc = max(exp(logspc) - log_offset, 0)."""
        F_._check(logspc, "WORLDVocoder")
        x = logspc.contiguous()
        y = torch.empty_like(x)
        N.call("v100_exp_clip", x, y, float(self.log_offset), x.numel())
        return y

    # ---- pyworld-owned analysis / synthesis ---------------------------------------------------------
    @staticmethod
    def _pyworld():
        try:
            import pyworld
        except ImportError as e:
            raise RuntimeError("WORLD analysis/synthesis (DIO, CheapTrick, D4C, synthesize) needs pyworld, which is "
                               "not installed; only the reference-owned glue runs without it") from e
        return pyworld

    def forward(self, waveform: torch.Tensor):
        return self.encode(waveform)

    def encode(self, waveform: torch.Tensor, f0_floor: float = 80.0, f0_ceil: float = 400.0):
        pyworld = self._pyworld()
        w = waveform.cpu().numpy().astype(np.double)
        f0, time_axis = pyworld.dio(w, self.sample_rate, f0_floor=f0_floor, f0_ceil=f0_ceil, frame_period=self.frame_period)
        spc = pyworld.cheaptrick(w, f0, time_axis, self.sample_rate, fft_size=self.n_fft)
        logspc = np.log(spc + self.log_offset)
        ap = pyworld.d4c(w, f0, time_axis, self.sample_rate, fft_size=self.n_fft)
        codeap = pyworld.code_aperiodicity(ap, self.sample_rate)
        feat = logspc @ self.sp2mc_matrix if self.use_mcep else logspc
        return (torch.from_numpy(f0.astype(np.float32)), torch.from_numpy(feat.astype(np.float32)),
                torch.from_numpy(codeap.astype(np.float32)))

    def decode(self, f0: torch.Tensor, logspc_or_mcep: torch.Tensor, codeap: torch.Tensor) -> np.ndarray:
        pyworld = self._pyworld()
        f0 = f0.cpu().numpy().astype(np.double, order="C")
        feat = logspc_or_mcep.cpu().numpy().astype(np.double)
        logspc = feat @ self.mc2sp_matrix if self.use_mcep else feat
        codeap = codeap.cpu().numpy().astype(np.double, order="C")
        spc = np.maximum(np.exp(logspc) - self.log_offset, 0).copy(order="C")
        ap = pyworld.decode_aperiodicity(codeap, self.sample_rate, self.n_fft)
        return pyworld.synthesize(f0, spc, ap, self.sample_rate, frame_period=self.frame_period)
