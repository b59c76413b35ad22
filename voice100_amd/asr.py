"""ASR side of the hot path: ConvVoiceEncoder, LinearCharDecoder, AudioToTextCTC.

Drop-in for voice100/models/asr.py:62-196 -- same class names, constructor
arguments, forward() layouts and state_dict keys -- with the convolution stacks
running on the MI355X kernels.  The model derives from Voice100ModelBase
(= pytorch_lightning.LightningModule when that is importable, else the shim in
voice100_amd/_base.py with the same hooks), so `Trainer.fit`, `load_from_checkpoint`
and the reference's ONNX exporter flow see a drop-in module; voice100_amd.trainer
is the step loop used where Lightning is absent.
"""
import torch
from torch import nn

from . import _stock
from . import functional as F_
from ._base import Voice100ModelBase, tracing
from .audio import BatchSpectrogramAugumentation
from .layers import InvertedResidual, PointwiseConv1d

__all__ = ["ConvVoiceEncoder", "LinearCharDecoder", "AudioToTextCTC"]


class ConvVoiceEncoder(nn.Module):
    """9 inverted-residual blocks, k = 11 (stride 2), 19, 27, 35, 51, 59, 67, 75, 83 (asr.py:62-82)."""

    def __init__(self, in_channels, out_channels, hidden_size):
        super().__init__()
        half = hidden_size // 2
        spec = [  # (cin, cout, k, stride, residual)
            (in_channels, half, 11, 2, False),
            (half, half, 19, 1, True), (half, half, 27, 1, True), (half, half, 35, 1, True),
            (half, hidden_size, 51, 1, False),
            (hidden_size, hidden_size, 59, 1, True), (hidden_size, hidden_size, 67, 1, True),
            (hidden_size, hidden_size, 75, 1, True),
            (hidden_size, out_channels, 83, 1, False),
        ]
        self.layers = nn.Sequential(*[
            InvertedResidual(ci, co, kernel_size=k, stride=s, use_residual=r) for ci, co, k, s, r in spec])

    def forward(self, embed: torch.Tensor) -> torch.Tensor:
        if self.training and not tracing():
            return F_.ir_stack_train(self.layers, embed)      # the nine blocks through the stack executor: one host call per direction
        return self.layers(embed)

    def output_length(self, embed_len: torch.Tensor) -> torch.Tensor:
        half = F_.half_length(embed_len)          # the augmentation pass already wrote (len + 1) // 2 next to the lengths
        if half is not None:
            return half
        return torch.div(embed_len + 1, 2, rounding_mode="trunc")


class LinearCharDecoder(nn.Module):
    """Dropout(0.2) then a biased 1x1 conv to the vocabulary (asr.py:85-94)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.layers = nn.Sequential(nn.Dropout(0.2), PointwiseConv1d(in_channels, out_channels, bias=True))
        self._keep_mask = None     # parity tests inject a fixed keep mask here

    def forward(self, enc_out: torch.Tensor) -> torch.Tensor:
        drop, conv = self.layers[0], self.layers[1]
        if tracing():
            return conv(drop(enc_out))
        if (self.training and 0.0 < drop.p < 1.0 and self._keep_mask is None and type(conv) is PointwiseConv1d
                and not (conv._forward_hooks or conv._forward_pre_hooks or drop._forward_hooks or drop._forward_pre_hooks
                         or conv._backward_hooks or drop._backward_hooks)):
            # one autograd node for the pair: the keep mask is applied in the data-gradient GEMM's epilogue on the way back
            return F_.dropout_pointwise_conv1d(enc_out, conv.weight, conv.bias, drop.p)
        x = F_.dropout(enc_out, drop.p, self.training, self._keep_mask)
        return conv(x)


class AudioToTextCTC(Voice100ModelBase):
    """audio [B, T, audio_size] fp32 -> logits [B, (T+1)//2, vocab_size] (asr.py:97-196)."""

    def __init__(self, audio_size, embed_size, vocab_size, hidden_size, learning_rate=0.001, weight_decay=0.00004):
        super().__init__()
        self.save_hyperparameters()
        self.embed_size = embed_size
        self.encoder = ConvVoiceEncoder(audio_size, embed_size, hidden_size)
        self.decoder = LinearCharDecoder(embed_size, vocab_size)
        self.criterion = nn.CTCLoss(zero_infinity=True)     # kept for API parity; the step uses functional.ctc_loss
        self.batch_augment = BatchSpectrogramAugumentation()
        self.batch_augment.emit_transposed = True     # forward() opens with transpose(1, 2): the augmentation pass writes that layout too
        self.do_normalize = False

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        if tracing():                            # torch.jit.trace / torch.onnx.export: plain aten ops (_stock.py)
            return torch.transpose(self.decoder(self.encoder(torch.transpose(audio, 1, 2))), 1, 2)
        x = F_.transpose_last2(audio)            # [B,T,C] -> [B,C,T]
        # The channel-major path runs under no_grad (its result is detached): take it only when nothing could ask for a gradient
        # through this forward -- autograd off, or neither the input nor any parameter requires one (an eval-mode model with
        # trainable parameters called with autograd ON is BatchNorm-frozen fine-tuning: the per-module path below is
        # differentiable like the reference's eval-mode BatchNorm) -- or at precision "fp16", which is inference-only by design.
        if not self.training and (not torch.is_grad_enabled() or F_.get_matmul_precision() == "fp16"
                                  or not (audio.requires_grad or any(p.requires_grad for p in self.parameters()))):
            y = self._forward_eval_cm(x)
            if y is not None:
                return y
        x = self.encoder(x)
        x = self.decoder(x)
        return F_.transpose_last2(x)             # [B,V,T'] -> [B,T',V]

    def _forward_eval_cm(self, x: torch.Tensor):
        """Inference at the 16-bit precisions: after the stride-2 opener the eight stride-1 blocks and the vocabulary head run on
        CHANNEL-MAJOR activations (functional.inverted_residual_eval_cm): one GEMM over the whole batch's columns per 1x1 convolution
        -- 1-second chunks (51 output frames) fill the 128-column tiles -- and packed short rows in the depthwise stage.  Same
        arithmetic per element as the per-module path (asr.py:62-94); None when the path does not apply (fp32, hooks, long rows ...)."""
        enc, dec = self.encoder, self.decoder
        layers = list(enc.layers)
        drop, conv = dec.layers[0], dec.layers[1]
        mods = (enc, enc.layers, dec, dec.layers, drop, conv)
        if enc.training or dec.training or any(m._forward_hooks or m._forward_pre_hooks for m in mods):
            return None
        B, _, T = x.shape
        T1 = int(enc.output_length(torch.tensor(T)))
        if layers[0].stride != 2 or T1 != F_.conv_out_len(T, int(layers[0].kernel_size), 2):
            return None
        if not F_.eval_cm_supported(layers[1:], T1, batch=B):
            return None
        with torch.no_grad():
            y = layers[0](x)                     # the stride-2 opener: batch-major (its depthwise kernel is the register-window one)
            xc = F_.bct_to_cm(y)
            xc = F_.ir_stack_eval_cm(layers[1:], xc, B, T1)      # the eight stride-1 blocks: one call into the library
            logits = F_.pointwise_conv1d_cm(xc, conv.weight, conv.bias)
            return F_.cm_to_btc(logits, B, T1)

    def output_length(self, audio_len: torch.Tensor) -> torch.Tensor:
        return self.encoder.output_length(audio_len)

    def normalize(self, audio: torch.Tensor, audio_len: torch.Tensor) -> torch.Tensor:
        """Masked per-utterance mean / std normalisation over time (asr.py:124-131).  Off by default in the reference
        (`do_normalize = False`, asr.py:108): a handful of reductions on an 8 MB tensor, stock torch ops."""
        from .tts import generate_padding_mask
        mask = torch.unsqueeze(generate_padding_mask(audio[:, :, 0], audio_len), dim=2)
        n = torch.sum(mask, dim=1, keepdim=True)
        mean = torch.sum(audio * mask, dim=1, keepdim=True) / n
        audio = (audio - mean) * mask
        std = torch.sqrt(torch.sum(audio ** 2, dim=1, keepdim=True) / n)
        return audio / (std + 1e-15) * mask

    def _calc_batch_loss(self, batch):
        (audio, audio_len), (text, text_len) = batch
        if self.training:
            audio, audio_len = self.batch_augment(audio, audio_len)
        if self.do_normalize:
            audio = self.normalize(audio, audio_len)
        logits = self.forward(audio)                         # [B, T', V]
        logits_len = self.output_length(audio_len)
        # log_softmax + CTCLoss(blank=0, mean, zero_infinity=True) fused in the HIP lattice kernels (K10)
        return F_.ctc_loss(logits, text, logits_len, text_len, blank=0)

    # LightningModule-style hooks the reference's trainers call (asr.py:154-178)
    def training_step(self, batch, batch_idx=0):
        loss = self._calc_batch_loss(batch)
        self.log_dict({"train_loss": loss})
        return loss

    def validation_step(self, batch, batch_idx=0):
        metrics = {"val_loss": self._calc_batch_loss(batch)}
        self.log_dict(metrics)
        return metrics

    def test_step(self, batch, batch_idx=0):
        metrics = {"test_loss": self._calc_batch_loss(batch)}
        self.log_dict(metrics)
        return metrics

    def configure_optimizers(self):
        # same update rule as the reference's torch.optim.Adam (asr.py:169-176); on the GPU PyTorch's single fused
        # multi-tensor kernel replaces the ~12 foreach launches per step
        params = list(self.parameters())
        if all(p.is_cuda for p in params):
            from .optim import FusedAdam                    # the same update as one launch for the whole model (csrc/adam.hip)
            optimizer = FusedAdam(params, lr=self.hparams.learning_rate, weight_decay=self.hparams.weight_decay)
        else:
            optimizer = torch.optim.Adam(params, lr=self.hparams.learning_rate, weight_decay=self.hparams.weight_decay)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=1, gamma=0.98)
        return {"optimizer": optimizer, "lr_scheduler": scheduler}

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("voice100.models.asr.AudioToTextCTC")
        parser.add_argument("--learning_rate", type=float, default=0.001)
        parser.add_argument("--weight_decay", type=float, default=0.00004)
        parser.add_argument("--hidden_size", type=float, default=512)    # sic: float in the reference (asr.py:185)
        parser.add_argument("--embed_size", type=float, default=512)
        return parent_parser

    @staticmethod
    def from_argparse_args(args, **kwargs):
        return AudioToTextCTC(embed_size=int(args.embed_size), hidden_size=int(args.hidden_size),
                              learning_rate=args.learning_rate, weight_decay=args.weight_decay, **kwargs)
