"""Conv front/back-ends of the reference's v2 models, state_dict-compatible.

Reference: voice100/models/_layers_v2.py:29-103 (ConvLayerBlock, ConvTransposeLayerBlock,
get_conv_layers), used by AudioToAlignText (`_asr_v2.py:31`, config/asr_en_base.yaml:16-18:
64->512 k5 s2 p2, 512->512 k5 s1 p2) and AlignTextToAudio (`_tts_v2.py:39`,
config/tts_en_base.yaml:20-23: 1024->512 k5, ConvTranspose 512->512 k5 s2 p2, 512->512 k5).

The sub-modules `conv` and `layer_norm` only own the parameters (keys `conv.weight`, `conv.bias`,
`layer_norm.weight`, `layer_norm.bias` as in the reference); the arithmetic runs on the HIP library:
dense conv = im2col + K1 MFMA GEMM, ConvTranspose1d = two tap-stacked K1 GEMMs, then one fused
channel-LayerNorm + GELU kernel.  GPU only, no fallback.  The LSTMs between these blocks stay on
PyTorch-ROCm (out of scope).
"""
from typing import List, Tuple

import torch
from torch import nn

from . import _stock
from . import functional as F_
from ._base import tracing

__all__ = ["ConvLayerBlock", "ConvTransposeLayerBlock", "get_conv_layers"]


class ConvLayerBlock(nn.Module):
    """Conv1d -> LayerNorm(out_channels) over channels -> GELU (_layers_v2.py:29-56). x [B, Cin, T] -> [B, Cout, T']."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int, padding: int, bias: bool) -> None:
        super().__init__()
        self.layer_norm = nn.LayerNorm(normalized_shape=out_channels)
        self.conv = nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                              stride=stride, padding=padding, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if tracing():
            return _stock.conv_layer_block(self, x, transpose=False)
        c = self.conv
        y = F_.conv1d_dense(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0])
        return F_.layer_norm_gelu(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)


class ConvTransposeLayerBlock(nn.Module):
    """ConvTranspose1d -> LayerNorm(out_channels) over channels -> GELU (_layers_v2.py:59-89).
    Built for the one configuration the reference uses: kernel_size=5, stride=2, padding=2 (output 2L-1)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int, padding: int, bias: bool) -> None:
        super().__init__()
        if (kernel_size, stride, padding) != (5, 2, 2):
            raise NotImplementedError("ConvTransposeLayerBlock: only kernel_size=5, stride=2, padding=2 is built "
                                      "(config/tts_en_base.yaml:22)")
        self.layer_norm = nn.LayerNorm(normalized_shape=out_channels)
        self.conv = nn.ConvTranspose1d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                                       stride=stride, padding=padding, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if tracing():
            return _stock.conv_layer_block(self, x, transpose=True)
        y = F_.conv_transpose1d_k5s2(x, self.conv.weight, self.conv.bias)
        return F_.layer_norm_gelu(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)


def get_conv_layers(in_channels: int, settings: List[Tuple]) -> nn.Module:
    """settings rows: (out_channels, transpose, kernel_size, stride, padding, bias) -- _layers_v2.py:92-106."""
    layers = []
    channels = in_channels
    for out_channels, transpose, kernel_size, stride, padding, bias in settings:
        cls = ConvTransposeLayerBlock if transpose else ConvLayerBlock
        layers.append(cls(channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias))
        channels = out_channels
    return nn.Sequential(*layers)
