"""Parameter-holding building blocks with the reference's state_dict layout.

Reference: voice100/models/asr.py:27-59 (ConvBNActivate, InvertedResidual).
The sub-modules here only OWN the parameters / buffers (so that
`conv.0.0.weight`, `conv.0.1.running_mean`, ... `conv.3.bias` load from a
reference checkpoint with strict=True); the arithmetic of a block runs as one
fused chain of HIP kernels (voice100_amd.functional), never through the
sub-modules' own forward().
"""
import warnings

import torch
from torch import nn

from . import _stock
from . import functional as F_
from ._base import tracing



class ConvBNActivate(nn.ModuleList):
    """Conv1d(bias=False) + BatchNorm1d + ReLU6 parameter group (asr.py:27-37).

    Index 0 = conv, 1 = batch norm, 2 = ReLU6 (no parameters), as in the reference's
    nn.Sequential, so the keys are `0.weight`, `1.weight`, `1.running_mean`, ...
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1):
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is never used by the reference networks")
        padding = (kernel_size - 1) // 2
        super().__init__([
            nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                      groups=groups, bias=False),
            nn.BatchNorm1d(out_channels),
            nn.ReLU6(inplace=True),
        ])

    def forward(self, x):
        raise RuntimeError("ConvBNActivate is a parameter group; run it through InvertedResidual "
                           "(fused HIP path) -- voice100_amd has no unfused fallback")


class InvertedResidual(nn.Module):
    """1x1 expand -> depthwise k -> 1x1 project, BatchNorm after each, ReLU6 after the first two,
    optional identity skip (asr.py:40-59).  x [B, Cin, T] fp32 on the GPU -> [B, Cout, T/stride]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, expand_ratio=4, use_residual=True):
        super().__init__()
        hidden_size = in_channels * expand_ratio
        self.use_residual = use_residual
        self.kernel_size = kernel_size
        self.stride = stride
        self.conv = nn.ModuleList([
            ConvBNActivate(in_channels, hidden_size, kernel_size=1),
            ConvBNActivate(hidden_size, hidden_size, kernel_size=kernel_size, stride=stride, groups=hidden_size),
            nn.Conv1d(hidden_size, out_channels, kernel_size=1, bias=False),
            nn.BatchNorm1d(out_channels),
        ])
        if use_residual and (in_channels != out_channels or stride != 1):
            raise ValueError("use_residual needs in_channels == out_channels and stride == 1")

    def train(self, mode: bool = True):
        # Any train()/eval() switch drops the eval-mode cache (folded BatchNorm coefficients, bf16 weights).  (The cache is keyed on
        # tensor versions, which the fused optimiser and the training forwards advance themselves -- functional._touched; this is the
        # belt to those braces.)
        self._eval_key = None
        return super().train(mode)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if tracing():                 # graph recording (torch.jit.trace / ONNX export): plain aten ops, see _stock.py
            return _stock.inverted_residual(self, x)
        pw, dw, pl, bn3 = self.conv[0], self.conv[1], self.conv[2], self.conv[3]
        bn1, bn2 = pw[1], dw[1]
        prec = F_.get_matmul_precision()
        if self.training:
            # the bf16 shadow the previous block wrote beside x (activation storage level 4): valid only for exactly this
            # version of x -- any in-place edit of x in between invalidates it
            sh = getattr(x, "_v100_shadow", None)
            x16 = sh[0] if (sh is not None and sh[1] == x._version and sh[0].shape[:2] == x.shape[:2]
                            and sh[0].shape[2] == F_.pitch16(x.shape[2], x.shape[0])) else None
            y, y16 = F_.InvertedResidualTrainFn.apply(
                x, pw[0].weight, bn1.weight, bn1.bias, dw[0].weight, bn2.weight, bn2.bias, pl.weight, bn3.weight, bn3.bias,
                bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                bn3.running_mean, bn3.running_var, bn3.num_batches_tracked,
                self.kernel_size, self.stride, self.use_residual, prec, F_.prepared_weights_of(self, prec), x16, True)
            if y16 is not None:
                y._v100_shadow = (y16, y._version)
            return y
        # eval mode.  Without autograd (torch.no_grad(), or nothing that requires grad): the inference kernels -- BatchNorm folded into
        # the three launches, 16-bit hidden tensors, cached coefficients.  WITH autograd the reference's block is an ordinary
        # differentiable function of x and of its parameters whose BatchNorms use their running statistics (partial-freeze fine-tuning:
        # block.eval() inside a training model; also every `model.eval(); model(x)` without no_grad, where the graph is built and
        # thrown away): that runs the training executor with FROZEN statistics -- same values as the inference kernels up to rounding,
        # gradients to x and to every parameter that requires one, nothing updated.  It is several times slower than the inference
        # kernels, hence the warning: wrap inference in torch.no_grad().
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if prec == "fp16":
                # the inference-only precision has no gradient kernels: an input whose gradient somebody asked for is refused, anything
                # else (model.eval(); model(x) without no_grad) gets the inference result, DETACHED, with a warning
                if x.requires_grad and x.is_leaf:
                    raise RuntimeError("InvertedResidual: precision 'fp16' is inference-only (no gradient kernels); call it under "
                                       "torch.no_grad(), or use 'bf16' / 'fp32'")
                warnings.warn("voice100_amd: eval-mode InvertedResidual at precision 'fp16' called with autograd on; 'fp16' is "
                              "inference-only, the output is DETACHED. Wrap inference in torch.no_grad().", stacklevel=2)
                with torch.no_grad():
                    return F_.inverted_residual_eval_cached(self, x.detach(), prec)
            warnings.warn("voice100_amd: eval-mode InvertedResidual called with autograd on: running the differentiable "
                          "frozen-statistics path (training kernels, several times slower than the inference kernels). "
                          "Wrap inference in torch.no_grad().", stacklevel=2)
            y, _ = F_.InvertedResidualTrainFn.apply(
                x, pw[0].weight, bn1.weight, bn1.bias, dw[0].weight, bn2.weight, bn2.bias, pl.weight, bn3.weight, bn3.bias,
                bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                bn3.running_mean, bn3.running_var, bn3.num_batches_tracked,
                self.kernel_size, self.stride, self.use_residual, prec, None, None, False, True)
            return y
        with torch.no_grad():
            return F_.inverted_residual_eval_cached(self, x.detach(), prec)


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) whose forward/backward run on the GEMM kernel (asr.py:91; tts.py:26,77)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(in_channels, out_channels, kernel_size=1, padding=0, bias=bias)

    def forward(self, x):
        if tracing():
            return _stock.pointwise_conv1d(self, x)
        return F_.pointwise_conv1d(x, self.weight, self.bias)
