"""TTS side of the hot path: VoiceDecoder, TextToAlignTextModel, AlignTextToAudioModel and the
WORLD feature norm / loss glue.  Drop-in for voice100/models/tts.py:13-262 and
voice100/models/_layers_v1.py:14-138 (same class names, constructor arguments, forward layouts and
state_dict keys); the convolution stacks, the transposed convolution and the predict epilogue run
on the MI355X kernels.
"""
from argparse import ArgumentParser
from typing import Tuple

import torch
from torch import nn

from . import _stock
from . import functional as F_
from ._base import Voice100ModelBase, tracing
from .layers import InvertedResidual, PointwiseConv1d

__all__ = ["VoiceDecoder", "TextToAlignTextModel", "AlignTextToAudioModel", "WORLDNorm", "WORLDLoss",
           "generate_padding_mask", "adjust_size"]


def generate_padding_mask(x: torch.Tensor, length: torch.Tensor) -> torch.Tensor:
    """[B, L] float mask, 1 where position < length (_layers_v1.py:14-24)."""
    assert x.dim() == 2
    assert length.dim() == 1
    return (torch.arange(x.shape[1], device=x.device)[None, :] < length[:, None]).to(x.dtype)


def adjust_size(x: torch.Tensor, y: torch.Tensor):
    """Truncate both to the shorter time axis (_layers_v1.py:27-34)."""
    n = min(x.shape[1], y.shape[1])
    return x[:, :n], y[:, :n]


class ConvTranspose1d(nn.ConvTranspose1d):
    """The one transposed convolution of the reference (k=5, stride=2, padding=2; tts.py:22) on the GEMM kernel."""

    def forward(self, x):
        if self.kernel_size != (5,) or self.stride != (2,) or self.padding != (2,) or self.output_padding != (0,):
            raise RuntimeError("only ConvTranspose1d(kernel_size=5, stride=2, padding=2) is built")
        if tracing():
            return _stock.conv_transpose1d(self, x)
        return F_.conv_transpose1d_k5s2(x, self.weight, self.bias)


class VoiceDecoder(nn.Module):
    """4 IR @H (k=65,33,17,11) -> ConvTranspose1d(H, H/2, k5 s2 p2) -> 3 IR @H/2 (k=33,11,7) -> 1x1 (tts.py:13-29).
    [B, H, L] -> [B, out_channels, 2L-1]."""

    def __init__(self, hidden_size, out_channels) -> None:
        super().__init__()
        half = hidden_size // 2
        self.layers = nn.Sequential(
            InvertedResidual(hidden_size, hidden_size, kernel_size=65),
            InvertedResidual(hidden_size, hidden_size, kernel_size=33),
            InvertedResidual(hidden_size, hidden_size, kernel_size=17),
            InvertedResidual(hidden_size, hidden_size, kernel_size=11),
            ConvTranspose1d(hidden_size, half, kernel_size=5, padding=2, stride=2),
            InvertedResidual(half, half, kernel_size=33),
            InvertedResidual(half, half, kernel_size=11),
            InvertedResidual(half, half, kernel_size=7),
            PointwiseConv1d(half, out_channels, bias=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and not tracing():
            # 4 blocks | transposed conv | 3 blocks | head: each run of blocks is one call into the stack executor
            x = F_.ir_stack_train(self.layers[0:4], x)
            x = self.layers[4](x)
            x = F_.ir_stack_train(self.layers[5:8], x)
            return self.layers[8](x)
        return self.layers(x)


class WORLDLoss(nn.Module):
    """Masked BCE (has-f0) + MSE/L1 (f0, log-spectrum with optional mel-slope weights, coded aperiodicity),
    each summed over valid frames / number of valid frames (_layers_v1.py:37-93).  On the GPU the four terms and their
    gradient come from ONE fused HIP pass (functional.world_loss, csrc/world_loss.hip); `forward` keeps the reference's
    signature (eight separate tensors), `fused` takes the decoder output and the raw targets as the model has them."""

    def __init__(self, loss: str = "mse", use_mel_weights: bool = False, sample_rate: int = 16000, n_fft: int = 512,
                 device=None, dtype=None) -> None:
        super().__init__()
        if loss not in ("l1", "mse"):
            raise ValueError("Unknown loss type")
        self.loss = loss
        if use_mel_weights:
            f = (sample_rate / n_fft) * torch.arange(n_fft // 2 + 1, device=device,
                                                    dtype=dtype if dtype is not None else torch.float32)
            dm = 1127 / (700 + f)
            self.register_buffer("logspc_weights", dm / torch.sum(dm), persistent=False)
        else:
            self.logspc_weights = None

    def _el(self, a, b):
        return (a - b) ** 2 if self.loss == "mse" else (a - b).abs()

    def fused(self, length, pred, f0, logspc, codeap, norm):
        """pred [B, T', 2+S+Cap] (the decoder output), raw targets, WORLDNorm `norm`: hasf0 = f0 >= 30, normalisation,
        adjust_size, masks, the four terms and d/dpred in one kernel (tts.py:203-211 + _layers_v1.py:68-93)."""
        nv = (norm.f0_mean, norm.f0_std, norm.logspc_mean, norm.logspc_std, norm.codeap_mean, norm.codeap_std)
        out = F_.world_loss(pred, length, None, f0, logspc, codeap, nv, self.logspc_weights, self.loss)
        return out[0], out[1], out[2], out[3]

    def forward(self, length, hasf0_logits, f0_hat, logspc_hat, codeap_hat, hasf0, f0, logspc, codeap):
        if hasf0_logits.is_cuda and not tracing():
            # the reference's signature on the fused kernel: the four predictions are packed into one [B, T', A] tensor
            # (one cat; autograd splits the gradient), targets are already normalised
            pred = torch.cat([hasf0_logits[:, :, None], f0_hat[:, :, None], logspc_hat, codeap_hat], dim=2)
            out = F_.world_loss(pred, length, hasf0, f0, logspc, codeap, None, self.logspc_weights, self.loss)
            return out[0], out[1], out[2], out[3]
        hasf0_logits, hasf0 = adjust_size(hasf0_logits, hasf0)
        f0_hat, f0 = adjust_size(f0_hat, f0)
        logspc_hat, logspc = adjust_size(logspc_hat, logspc)
        codeap_hat, codeap = adjust_size(codeap_hat, codeap)
        mask = generate_padding_mask(f0, length)
        hasf0_loss = nn.functional.binary_cross_entropy_with_logits(hasf0_logits, hasf0, reduction="none") * mask
        f0_loss = self._el(f0_hat, f0) * hasf0 * mask
        if self.logspc_weights is not None:
            logspc_loss = torch.sum(self._el(logspc_hat, logspc) * self.logspc_weights[None, None, :], dim=2) * mask
        else:
            logspc_loss = torch.mean(self._el(logspc_hat, logspc), dim=2) * mask
        codeap_loss = torch.mean(self._el(codeap_hat, codeap), dim=2) * mask
        ms = torch.sum(mask)
        return (torch.sum(hasf0_loss) / ms, torch.sum(f0_loss) / ms, torch.sum(logspc_loss) / ms, torch.sum(codeap_loss) / ms)


class WORLDNorm(nn.Module):
    """Frozen per-feature mean / std (_layers_v1.py:96-138); keys f0_/logspc_/codeap_ x mean/std."""

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
        return ((f0 - self.f0_mean) / self.f0_std, (mcep - self.logspc_mean) / self.logspc_std,
                (codeap - self.codeap_mean) / self.codeap_std)

    @torch.no_grad()
    def unnormalize(self, f0, mcep, codeap):
        return (self.f0_std * f0 + self.f0_mean, self.logspc_std * mcep + self.logspc_mean,
                self.codeap_std * codeap + self.codeap_mean)


class TextToAlignTextModel(Voice100ModelBase):
    """text [B, L] int64 -> [B, L, 2] log(gap+1), log(len+1) predictions (tts.py:67-149)."""

    def __init__(self, vocab_size, hidden_size, learning_rate=1e-3) -> None:
        super().__init__()
        self.save_hyperparameters()
        self.embedding = nn.Embedding(vocab_size, hidden_size)
        self.layers = nn.Sequential(
            InvertedResidual(hidden_size, hidden_size, kernel_size=5),
            InvertedResidual(hidden_size, hidden_size, kernel_size=11),
            InvertedResidual(hidden_size, hidden_size, kernel_size=17),
            InvertedResidual(hidden_size, hidden_size, kernel_size=29),
            PointwiseConv1d(hidden_size, 2, bias=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if tracing():
            return torch.transpose(self.layers(_stock.embedding_bct(x, self.embedding.weight)), 1, 2)
        x = F_.embedding_bct(x, self.embedding.weight)          # [B, H, L]
        if self.training:
            x = self.layers[4](F_.ir_stack_train(self.layers[0:4], x))
        else:
            x = self.layers(x)
        return F_.transpose_last2(x)                             # [B, L, 2]

    def align(self, text: torch.Tensor, align: torch.Tensor, head=5, tail=5) -> torch.Tensor:
        """Expand text by predicted (gap, length) pairs -- integer, host side (tts.py:89-110)."""
        assert text.dim() == 1
        assert align.dim() == 2
        text, align = text.cpu(), align.cpu()
        aligntext = torch.zeros(head + int(torch.sum(align)) + tail, dtype=text.dtype)
        t = head
        for i in range(align.shape[0]):
            t += align[i, 0].item()
            s = round(t)
            t += align[i, 1].item()
            e = round(t)
            if s == e:
                e = max(0, e + 1)
            aligntext[s:e] = text[i]
        return aligntext

    def _calc_batch_loss(self, batch) -> torch.Tensor:
        (text, text_len), (align, align_len) = batch
        align = align[:, :-1].reshape([align.shape[0], -1, 2])
        pred = self.forward(text)
        logalign = torch.log((align + 1).to(pred.dtype))
        loss = torch.mean(torch.abs(logalign - pred), dim=2)
        mask = generate_padding_mask(text, text_len)
        return torch.sum(loss * mask) / torch.sum(mask)

    def training_step(self, batch, batch_idx=0):
        loss = self._calc_batch_loss(batch)
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx=0):
        loss = self._calc_batch_loss(batch)
        self.log("val_loss", loss)
        return {"val_loss": loss}

    def configure_optimizers(self):
        params = [p for p in self.parameters() if p.requires_grad]
        if all(p.is_cuda for p in params):
            from .optim import FusedAdam
            return FusedAdam(params, lr=self.hparams.learning_rate)
        return torch.optim.Adam(params, lr=self.hparams.learning_rate)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--hidden_size", type=int, default=512)
        parser.add_argument("--learning_rate", type=float, default=1e-3)
        return parser

    @staticmethod
    def from_argparse_args(args, **kwargs):
        return TextToAlignTextModel(hidden_size=args.hidden_size, learning_rate=args.learning_rate, **kwargs)


class AlignTextToAudioModel(Voice100ModelBase):
    """aligned text [B, L] int64 -> WORLD features at 2L-1 frames (tts.py:152-262)."""

    def __init__(self, vocab_size: int, hidden_size: int, learning_rate: float = 1e-3, use_mcep: bool = False) -> None:
        super().__init__()
        self.save_hyperparameters()
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.sample_rate = 16000
        self.n_fft = 512
        self.hasf0_size = 1
        self.f0_size = 1
        self.logspc_size = 25 if use_mcep else self.n_fft // 2 + 1
        self.codeap_size = 1
        self.embedding = nn.Embedding(vocab_size, hidden_size)
        self.audio_size = self.hasf0_size + self.f0_size + self.logspc_size + self.codeap_size
        self.decoder = VoiceDecoder(hidden_size, self.audio_size)
        self.norm = WORLDNorm(self.logspc_size, self.codeap_size)
        self.criterion = WORLDLoss(use_mel_weights=not use_mcep, sample_rate=self.sample_rate, n_fft=self.n_fft)

    def _decode(self, aligntext: torch.Tensor) -> torch.Tensor:
        if tracing():
            return torch.transpose(self.decoder(_stock.embedding_bct(aligntext, self.embedding.weight)), 1, 2)
        x = F_.embedding_bct(aligntext, self.embedding.weight)   # [B, H, L]
        x = self.decoder(x)                                      # [B, A, 2L-1]
        return F_.transpose_last2(x)                             # [B, 2L-1, A]

    def forward(self, aligntext: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        x = self._decode(aligntext)
        hasf0_logits, f0_hat, logspc_hat, codeap_hat = torch.split(
            x, [self.hasf0_size, self.f0_size, self.logspc_size, self.codeap_size], dim=2)
        return hasf0_logits[:, :, 0], f0_hat[:, :, 0], logspc_hat, codeap_hat

    @torch.no_grad()
    def predict(self, aligntext: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        x = self._decode(aligntext)
        n = self.norm
        if tracing():
            return _stock.world_unnormalize_gate(x, n, [self.hasf0_size, self.f0_size, self.logspc_size, self.codeap_size])
        return F_.world_unnormalize_gate(x, n.f0_mean, n.f0_std, n.logspc_mean, n.logspc_std, n.codeap_mean, n.codeap_std)

    def _calc_batch_loss(self, batch):
        (f0, f0_len, logspc, codeap), (aligntext, aligntext_len) = batch
        # tts.py:203-211 -- hasf0 = f0 >= 30, WORLDNorm.normalize, forward, WORLDLoss -- with everything after the decoder
        # in one fused kernel over the decoder output (no split / normalise / mask / reduce passes)
        return self.criterion.fused(f0_len, self._decode(aligntext), f0, logspc, codeap, self.norm)

    def _step(self, task: str, batch) -> torch.Tensor:
        hasf0_loss, f0_loss, logspc_loss, codeap_loss = self._calc_batch_loss(batch)
        loss = hasf0_loss + f0_loss + logspc_loss + codeap_loss
        self.log(f"{task}_loss", loss)                           # tts.py:232-237
        self.log(f"{task}_hasf0_loss", hasf0_loss)
        self.log(f"{task}_f0_loss", f0_loss)
        self.log(f"{task}_logspc_loss", logspc_loss)
        self.log(f"{task}_codeap_loss", codeap_loss)
        return loss

    def training_step(self, batch, batch_idx=0) -> torch.Tensor:
        return self._step("train", batch)

    def validation_step(self, batch, batch_idx=0):
        return {"val_loss": self._step("val", batch)}

    def test_step(self, batch, batch_idx=0):
        return {"test_loss": self._step("test", batch)}

    def configure_optimizers(self):
        params = [p for p in self.parameters() if p.requires_grad]
        if all(p.is_cuda for p in params):
            from .optim import FusedAdam
            return FusedAdam(params, lr=self.hparams.learning_rate)
        return torch.optim.Adam(params, lr=self.hparams.learning_rate)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--hidden_size", type=int, default=512)
        parser.add_argument("--audio_stat", type=str)
        parser.add_argument("--learning_rate", type=float, default=1e-3)
        return parser

    @staticmethod
    def from_argparse_args(args, **kwargs):
        model = AlignTextToAudioModel(hidden_size=args.hidden_size, learning_rate=args.learning_rate,
                                      use_mcep=args.vocoder == "world_mcep", **kwargs)
        if not args.resume_from_checkpoint:
            if args.audio_stat is None:
                args.audio_stat = f"./data/{args.dataset}-stat.pt"
            model.norm.load_state_dict(torch.load(args.audio_stat))
        return model
