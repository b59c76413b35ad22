"""Stock-op forward of the hot-path modules, used ONLY while a graph is being recorded
(torch.jit.trace / torch.onnx.export; `_base.tracing()`).

The HIP kernels are opaque to a tracer, so the exporter of the reference
(voice100/export_onnx_v1.py:35-57, 60-84, 96-125) would see nothing it can lower.  While tracing, each
module therefore runs the plain aten ops the reference's own nn.Sequential would run on the module's own
parameters, which gives the same graph the reference produces.  This is not a compute fallback: outside
a trace, CPU tensors or a missing extension still raise (functional._check / _native.load).
"""
import torch
import torch.nn.functional as F


def conv_bn_act(group, x: torch.Tensor, training: bool) -> torch.Tensor:
    """ConvBNActivate (asr.py:27-37): group[0] conv, group[1] batch norm, then ReLU6."""
    conv, bn = group[0], group[1]
    x = F.conv1d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    x = batch_norm(bn, x, training)
    return F.relu6(x)


def batch_norm(bn, x: torch.Tensor, training: bool) -> torch.Tensor:
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)


def inverted_residual(blk, x: torch.Tensor) -> torch.Tensor:
    """InvertedResidual.forward (asr.py:55-59)."""
    t = blk.training
    h = conv_bn_act(blk.conv[0], x, t)
    h = conv_bn_act(blk.conv[1], h, t)
    pl = blk.conv[2]
    h = F.conv1d(h, pl.weight, None, pl.stride, pl.padding)
    h = batch_norm(blk.conv[3], h, t)
    return x + h if blk.use_residual else h


def pointwise_conv1d(conv, x: torch.Tensor) -> torch.Tensor:
    return F.conv1d(x, conv.weight, conv.bias)


def conv_transpose1d(conv, x: torch.Tensor) -> torch.Tensor:
    return F.conv_transpose1d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.output_padding)


def embedding_bct(idx: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    return torch.transpose(F.embedding(idx, table), 1, 2)


def world_unnormalize_gate(x, norm, sizes):
    """AlignTextToAudioModel.predict's epilogue (tts.py:192-201): split, un-normalise, zero f0 where the gate is negative."""
    hasf0_logits, f0_hat, logspc_hat, codeap_hat = torch.split(x, sizes, dim=2)
    hasf0_logits, f0_hat = hasf0_logits[:, :, 0], f0_hat[:, :, 0]
    f0 = norm.f0_std * f0_hat + norm.f0_mean
    logspc = norm.logspc_std * logspc_hat + norm.logspc_mean
    codeap = norm.codeap_std * codeap_hat + norm.codeap_mean
    f0 = torch.where(hasf0_logits < 0, torch.zeros(size=(1,), dtype=f0.dtype, device=f0.device), f0)
    return f0, logspc, codeap


def conv_layer_block(blk, x: torch.Tensor, transpose: bool) -> torch.Tensor:
    """ConvLayerBlock / ConvTransposeLayerBlock (_layers_v2.py:50-56, 83-89)."""
    c = blk.conv
    if transpose:
        x = F.conv_transpose1d(x, c.weight, c.bias, c.stride, c.padding, c.output_padding)
    else:
        x = F.conv1d(x, c.weight, c.bias, c.stride, c.padding)
    ln = blk.layer_norm
    x = F.layer_norm(torch.transpose(x, -2, -1), ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    return F.gelu(torch.transpose(x, -2, -1))
