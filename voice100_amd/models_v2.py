"""The reference's v2 models on the MI355X conv blocks (widened to the modules around them).

Reference: voice100/models/_asr_v2.py:18-121 (AudioToAlignText), voice100/models/_tts_v2.py:13-178
(AlignTextToAudio), voice100/models/_layers_v2.py:109-206 (WORLDLoss, WORLDNorm).  Same constructor arguments,
forward() layouts and state_dict keys, so the released v2 checkpoints (README.md:297-308) load with strict=True.

What runs where: the conv front/back-ends (`encoder` / `decoder`, get_conv_layers) run on the HIP library (dense conv as
im2col + MFMA GEMM, ConvTranspose, fused channel-LayerNorm + GELU), the CTC loss on the fused K10 kernel, the WORLD
un-normalisation / gating on the GPU; the bidirectional LSTMs, the embedding and the two Linear heads stay stock
PyTorch-ROCm modules (recurrent layers are outside the hot path).  GPU tensors only.
"""
from typing import List, Optional, Tuple

import torch
from torch import nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import functional as F_
from .audio import BatchSpectrogramAugumentation
from .layers_v2 import get_conv_layers

__all__ = ["AudioToAlignText", "AlignTextToAudio", "WORLDLoss", "WORLDNorm"]


def generate_padding_mask(x: torch.Tensor, length: torch.Tensor) -> torch.Tensor:
    """[B, L] float mask of positions < length (_layers_v2.py:17-26)."""
    assert x.dim() == 2 and length.dim() == 1
    return (torch.arange(x.shape[1], device=x.device)[None, :] < length[:, None].to(x.device)).to(x.dtype)


def adjust_size(x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Truncate the longer of the two along dim 1 (_layers_v2.py:109-116)."""
    if x.shape[1] > y.shape[1]:
        return x[:, :y.shape[1]], y
    if x.shape[1] < y.shape[1]:
        return x, y[:, :x.shape[1]]
    return x, y


class WORLDLoss(nn.Module):
    """Masked BCE (hasf0, hascodeap) + MSE/L1 (f0, logspc, codeap), each sum / sum(mask) (_layers_v2.py:119-161)."""

    def __init__(self, loss: str = "mse") -> None:
        super().__init__()
        self.hasf0_criterion = nn.BCEWithLogitsLoss(reduction="none")
        self.hascodeap_criterion = nn.BCEWithLogitsLoss(reduction="none")
        if loss == "l1":
            make = nn.L1Loss
        elif loss == "mse":
            make = nn.MSELoss
        else:
            raise ValueError("Unknown loss type")
        self.f0_criterion, self.logspc_criterion, self.codeap_criterion = make(reduction="none"), make(reduction="none"), make(reduction="none")

    def forward(self, length, hasf0_logits, f0_hat, logspc_hat, hascodeap_logits, codeap_hat, hasf0, f0, logspc, hascodeap, codeap):
        hasf0_logits, hasf0 = adjust_size(hasf0_logits, hasf0)
        f0_hat, f0 = adjust_size(f0_hat, f0)
        logspc_hat, logspc = adjust_size(logspc_hat, logspc)
        hascodeap_logits, hascodeap = adjust_size(hascodeap_logits, hascodeap)
        codeap_hat, codeap = adjust_size(codeap_hat, codeap)
        mask = generate_padding_mask(f0, length)
        hasf0_loss = self.hasf0_criterion(hasf0_logits, hasf0) * mask
        f0_loss = self.f0_criterion(f0_hat, f0) * hasf0 * mask
        logspc_loss = torch.mean(self.logspc_criterion(logspc_hat, logspc), axis=2) * mask
        hascodeap_loss = torch.mean(self.hascodeap_criterion(hascodeap_logits, hascodeap), axis=2) * mask
        codeap_loss = torch.mean(self.codeap_criterion(codeap_hat, codeap) * hascodeap, axis=2) * mask
        mask_sum = torch.sum(mask)
        return (torch.sum(hasf0_loss) / mask_sum, torch.sum(f0_loss) / mask_sum, torch.sum(logspc_loss) / mask_sum,
                torch.sum(hascodeap_loss) / mask_sum, torch.sum(codeap_loss) / mask_sum)


class WORLDNorm(nn.Module):
    """Per-feature mean/std of the WORLD features as frozen parameters (_layers_v2.py:164-206)."""

    def __init__(self, logspc_size: int, codeap_size: int, device=None, dtype=None):
        kw = {"device": device, "dtype": dtype}
        super().__init__()
        self.f0_std = nn.Parameter(torch.ones([1], **kw), requires_grad=False)
        self.f0_mean = nn.Parameter(torch.zeros([1], **kw), requires_grad=False)
        self.logspc_std = nn.Parameter(torch.ones([logspc_size], **kw), requires_grad=False)
        self.logspc_mean = nn.Parameter(torch.zeros([logspc_size], **kw), requires_grad=False)
        self.codeap_std = nn.Parameter(torch.ones([codeap_size], **kw), requires_grad=False)
        self.codeap_mean = nn.Parameter(torch.zeros([codeap_size], **kw), requires_grad=False)

    def forward(self, f0, mcep, codeap):
        return self.normalize(f0, mcep, codeap)

    @torch.no_grad()
    def normalize(self, f0, mcep, codeap):
        return (f0 - self.f0_mean) / self.f0_std, (mcep - self.logspc_mean) / self.logspc_std, (codeap - self.codeap_mean) / self.codeap_std

    @torch.no_grad()
    def unnormalize(self, f0, mcep, codeap):
        return self.f0_std * f0 + self.f0_mean, self.logspc_std * mcep + self.logspc_mean, self.codeap_std * codeap + self.codeap_mean


def _require_cuda(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: voice100_amd runs on the GPU only (no CPU fallback)")


class AudioToAlignText(nn.Module):
    """Conv encoder (stride 2) -> bidirectional LSTM -> Linear; CTC loss (_asr_v2.py:18-121).
    audio [B, T, audio_size], audio_len [B] -> (logits [T', B, vocab], logits_len [B] on the CPU)."""

    def __init__(self, audio_size: int, encoder_settings: List[List], decoder_num_layers: int, decoder_hidden_size: int,
                 vocab_size: int, learning_rate: float = 0.001) -> None:
        super().__init__()
        from argparse import Namespace
        self.hparams = Namespace(audio_size=audio_size, encoder_settings=encoder_settings, decoder_num_layers=decoder_num_layers,
                                 decoder_hidden_size=decoder_hidden_size, vocab_size=vocab_size, learning_rate=learning_rate)
        self.encoder = get_conv_layers(audio_size, encoder_settings)
        self.lstm = nn.LSTM(input_size=decoder_hidden_size, hidden_size=decoder_hidden_size, num_layers=decoder_num_layers,
                            dropout=0.2, bidirectional=True)
        self.dense = nn.Linear(decoder_hidden_size * 2, vocab_size)
        self.batch_augment = BatchSpectrogramAugumentation()

    def forward(self, audio: torch.Tensor, audio_len: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _require_cuda(audio, "AudioToAlignText")
        x = F_.transpose_last2(audio)                                  # [B, audio_size, T]
        x = self.encoder(x)
        x_len = torch.divide(audio_len + 1, 2, rounding_mode="trunc")
        x = F_.transpose_last2(x)                                      # [B, T', H]
        packed = pack_padded_sequence(x, x_len.cpu(), batch_first=True, enforce_sorted=False)
        packed_out, _ = self.lstm(packed)
        lstm_out, lstm_out_len = pad_packed_sequence(packed_out, batch_first=False)
        return self.dense(lstm_out), lstm_out_len

    def _calc_batch_loss(self, batch):
        (audio, audio_len), (text, text_len) = batch
        if self.training:
            audio, audio_len = self.batch_augment(audio, audio_len)
        logits, logits_len = self.forward(audio, audio_len)            # [T', B, V]
        # log_softmax + nn.CTCLoss(zero_infinity=True), mean reduction (_asr_v2.py:59-61): the fused K10 kernel
        return F_.ctc_loss(logits.transpose(0, 1).contiguous(), text, logits_len.to(logits.device), text_len.to(logits.device), blank=0)

    def training_step(self, batch, batch_idx=0):
        return self._calc_batch_loss(batch)

    def validation_step(self, batch, batch_idx=0):
        return {"val_loss": self._calc_batch_loss(batch)}

    def test_step(self, batch, batch_idx=0):
        return {"test_loss": self._calc_batch_loss(batch)}

    def configure_optimizers(self):
        params = list(self.parameters())
        return torch.optim.Adam(params, lr=self.hparams.learning_rate, fused=all(p.is_cuda for p in params))

    @torch.no_grad()
    def ctc_best_path(self, audio=None, audio_len=None, text=None, text_len=None, logits=None):
        """Greedy ids [T', B] when `text` is None, else the banded Viterbi alignment of _asr_v2.py:89-121 on the device
        kernel: (score [B], hist [B, T'] positions in the blank-expanded text, path [B, T'] labels, logits_len), hist / path
        zero past logits_len as the reference's pad_sequence leaves them.  (The reference overwrites `score` with the last
        utterance's path, _asr_v2.py:118 - here it is the per-utterance best score.)"""
        from .decode import ctc_best_path as device_best_path
        if logits is None:
            logits, logits_len = self.forward(audio, audio_len)
            logits = torch.log_softmax(logits, dim=-1)
        else:
            logits_len = audio_len
        if text is None:
            return logits.argmax(axis=-1)
        logits_len = logits_len.cpu()
        text_len = torch.minimum(logits_len, text_len.cpu())
        score, hist, path = device_best_path(logits.transpose(0, 1).contiguous(), text, logits_len, text_len)
        tmax = int(logits_len.max())
        live = torch.arange(tmax, device=hist.device)[None, :] < logits_len.to(hist.device)[:, None]
        return score, hist[:, :tmax] * live, path[:, :tmax] * live, logits_len


class AlignTextToAudio(nn.Module):
    """Embedding -> bidirectional LSTM -> conv decoder (x2 upsampling ConvTranspose) -> Linear to the WORLD features
    (_tts_v2.py:13-178).  aligntext [B, L] int64, aligntext_len [B] ->
    (hasf0_logits [B, T], f0_hat [B, T], logspc_hat [B, T, S], hascodeap_logits [B, T, Ca], codeap_hat [B, T, Ca]), T = 2L-1."""

    def __init__(self, vocab_size: int, logspc_size: int, codeap_size: int, encoder_num_layers: int, encoder_hidden_size: int,
                 decoder_settings: List[List], logspc_weight: float = 5.0, learning_rate: float = 1e-3, f0_size: int = 1,
                 audio_stat: Optional[str] = None) -> None:
        super().__init__()
        from argparse import Namespace
        self.hparams = Namespace(vocab_size=vocab_size, logspc_size=logspc_size, codeap_size=codeap_size,
                                 encoder_num_layers=encoder_num_layers, encoder_hidden_size=encoder_hidden_size,
                                 decoder_settings=decoder_settings, logspc_weight=logspc_weight, learning_rate=learning_rate,
                                 f0_size=f0_size, audio_stat=audio_stat)
        self.encoder_hidden_size = encoder_hidden_size
        self.vocab_size, self.f0_size, self.logspc_size, self.codeap_size = vocab_size, f0_size, logspc_size, codeap_size
        self.audio_size = 2 * f0_size + logspc_size + 2 * codeap_size
        self.embedding = nn.Embedding(vocab_size, encoder_hidden_size)
        self.lstm = nn.LSTM(input_size=encoder_hidden_size, hidden_size=encoder_hidden_size, num_layers=encoder_num_layers,
                            dropout=0.2, bidirectional=True)
        self.decoder = get_conv_layers(2 * encoder_hidden_size, decoder_settings)
        self.projection = nn.Linear(decoder_settings[-1][0], self.audio_size)
        self.norm = WORLDNorm(logspc_size, codeap_size)
        self.criterion = WORLDLoss()
        self.logspc_weight = logspc_weight
        if audio_stat is not None:
            self.norm.load_state_dict(torch.load(audio_stat))

    def forward(self, aligntext: torch.Tensor, aligntext_len: torch.Tensor):
        _require_cuda(aligntext, "AlignTextToAudio")
        x = self.embedding(aligntext)                                   # [B, L, H]
        packed = pack_padded_sequence(x, aligntext_len.cpu(), batch_first=True, enforce_sorted=False)
        packed_out, _ = self.lstm(packed)
        lstm_out, _ = pad_packed_sequence(packed_out, batch_first=True)
        x = F_.transpose_last2(lstm_out.contiguous())                   # [B, 2H, L]
        x = self.decoder(x)
        x = F_.transpose_last2(x)                                       # [B, T, C]
        x = self.projection(x)
        hasf0_logits, f0_hat, logspc_hat, hascodeap_logits, codeap_hat = torch.split(
            x, [self.f0_size, self.f0_size, self.logspc_size, self.codeap_size, self.codeap_size], dim=2)
        return hasf0_logits[:, :, 0], f0_hat[:, :, 0], logspc_hat, hascodeap_logits, codeap_hat

    def predict(self, aligntext: torch.Tensor, aligntext_len: torch.Tensor):
        hasf0, f0, logspc, hascodeap, codeap = self.forward(aligntext, aligntext_len)
        f0, logspc, codeap = self.norm.unnormalize(f0, logspc, codeap)
        f0 = torch.where(hasf0 < 0, torch.zeros(size=(1,), dtype=f0.dtype, device=f0.device), f0)
        codeap = torch.where(hascodeap < 0, torch.zeros(size=(1, 1), dtype=codeap.dtype, device=codeap.device), codeap)
        return f0, logspc, codeap

    def _calc_batch_loss(self, batch):
        (f0, f0_len, logspc, codeap), (aligntext, aligntext_len) = batch
        hasf0 = (f0 >= 30.0).to(torch.float32)
        hascodeap = (codeap < -0.2).to(torch.float32)
        f0, logspc, codeap = self.norm.normalize(f0, logspc, codeap)
        hasf0_logits, f0_hat, logspc_hat, hascodeap_logits, codeap_hat = self.forward(aligntext, aligntext_len)
        return self.criterion(f0_len, hasf0_logits, f0_hat, logspc_hat, hascodeap_logits, codeap_hat, hasf0, f0, logspc, hascodeap, codeap)

    def _total(self, losses):
        hasf0_loss, f0_loss, logspc_loss, hascodeap_loss, codeap_loss = losses
        return hasf0_loss + f0_loss + logspc_loss * self.logspc_weight + hascodeap_loss + codeap_loss

    def training_step(self, batch, batch_idx=0) -> torch.Tensor:
        return self._total(self._calc_batch_loss(batch))

    def validation_step(self, batch, batch_idx=0):
        return {"val_loss": self._total(self._calc_batch_loss(batch))}

    def test_step(self, batch, batch_idx=0):
        return {"test_loss": self._total(self._calc_batch_loss(batch))}

    def configure_optimizers(self):
        params = [p for p in self.parameters() if p.requires_grad]
        return torch.optim.Adam(params, lr=self.hparams.learning_rate, fused=all(p.is_cuda for p in params))
