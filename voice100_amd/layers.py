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
        # Any train()/eval() switch drops the eval-mode cache (folded BatchNorm coefficients, bf16 weights): training
        # updates parameters and running statistics through raw pointers / fused optimiser kernels, which do not bump
        # the tensor versions the cache is keyed on.
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
                            and sh[0].shape[2] == ((x.shape[2] + 7) & ~7)) else None
            y, y16 = F_.InvertedResidualTrainFn.apply(
                x, pw[0].weight, bn1.weight, bn1.bias, dw[0].weight, bn2.weight, bn2.bias, pl.weight, bn3.weight, bn3.bias,
                bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                bn3.running_mean, bn3.running_var, bn3.num_batches_tracked,
                self.kernel_size, self.stride, self.use_residual, prec, F_.prepared_weights_of(self, prec), x16, True)
            if y16 is not None:
                y._v100_shadow = (y16, y._version)
            return y
        # eval mode: frozen statistics, inference only (autograd through eval-mode BN is not on the reference's training path
        # and is not built).  The reference's inference scripts call model.eval(); model(x) without torch.no_grad(), and the
        # embedding-fed models hand this block an x that requires grad (the table is a Parameter): that must keep working, so
        # the result is returned detached.  Only an x the USER marked requires_grad (a leaf: somebody wants d/dx) is refused,
        # instead of silently handing back a gradient-less constant.
        if torch.is_grad_enabled() and x.requires_grad:
            if x.is_leaf:
                raise RuntimeError("InvertedResidual in eval mode is inference-only (frozen-BatchNorm fine-tuning is not "
                                   "built): call it under torch.no_grad(), or switch the block to train()")
            # (python's default warning filter shows this once per call site; every call warns so that a filter of "always" sees all)
            warnings.warn("voice100_amd: eval-mode InvertedResidual called with autograd on; its output is DETACHED "
                          "(inference only: nothing upstream of this block receives a gradient). Wrap inference in "
                          "torch.no_grad() to silence this.", stacklevel=2)
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
