"""Functional CPU restatement of the reference's inverted-residual CNN path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Everything here is a pure
function over a flat ``state`` dict whose keys are the reference's
``state_dict`` key names, so a reference checkpoint (or a golden fixture) can
be fed in unchanged.  Built on stock ``torch.nn.functional`` CPU ops in fp32,
so autograd supplies the gradients the HIP backward kernels are checked
against.

Reference anchors (relative to /root/reference):
  ConvBNActivate          voice100/models/asr.py:27-37
  InvertedResidual        voice100/models/asr.py:40-59
  ConvVoiceEncoder        voice100/models/asr.py:62-82
  LinearCharDecoder       voice100/models/asr.py:85-94
  AudioToTextCTC          voice100/models/asr.py:97-152
  VoiceDecoder            voice100/models/tts.py:13-29
  TextToAlignTextModel    voice100/models/tts.py:67-130
  AlignTextToAudioModel   voice100/models/tts.py:152-213
  WORLDLoss / WORLDNorm   voice100/models/_layers_v1.py:37-138
  ConvLayerBlock / ConvTransposeLayerBlock / get_conv_layers   voice100/models/_layers_v2.py:29-106
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5          # nn.BatchNorm1d default, asr.py:36
BN_MOMENTUM = 0.1

# (in_is_half, out_is_half?, kernel, stride, residual) tables ---------------
# asr.py:67-76: 9 blocks; channels are derived from (in_channels, hidden, out).
ENCODER_KERNELS = (11, 19, 27, 35, 51, 59, 67, 75, 83)
# tts.py:18-25
DECODER_KERNELS_A = (65, 33, 17, 11)
DECODER_KERNELS_B = (33, 11, 7)
# tts.py:73-76
ALIGN_KERNELS = (5, 11, 17, 29)


def encoder_block_specs(in_channels: int, out_channels: int, hidden_size: int):
    """(cin, cout, k, stride, residual) for ConvVoiceEncoder (asr.py:64-76)."""
    half = hidden_size // 2
    return [
        (in_channels, half, 11, 2, False),
        (half, half, 19, 1, True),
        (half, half, 27, 1, True),
        (half, half, 35, 1, True),
        (half, hidden_size, 51, 1, False),
        (hidden_size, hidden_size, 59, 1, True),
        (hidden_size, hidden_size, 67, 1, True),
        (hidden_size, hidden_size, 75, 1, True),
        (hidden_size, out_channels, 83, 1, False),
    ]


class BNUpdates(dict):
    """Collects the running-stat values a train-mode pass would write back."""


def _batch_norm(x: torch.Tensor, state: State, prefix: str, training: bool,
                updates: Optional[BNUpdates]) -> torch.Tensor:
    """BatchNorm1d over (B, T) per channel (asr.py:36).

    train: normalise with the biased batch variance; running_var gets the
    unbiased one, momentum 0.1; num_batches_tracked += 1.
    eval: normalise with running stats.
    """
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    rm, rv = state[prefix + ".running_mean"], state[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    n = x.shape[0] * x.shape[2]
    mean = x.mean(dim=(0, 2))
    var = x.var(dim=(0, 2), unbiased=False)
    y = (x - mean[None, :, None]) * torch.rsqrt(var + BN_EPS)[None, :, None]
    y = y * w[None, :, None] + b[None, :, None]
    if updates is not None:
        with torch.no_grad():
            unbiased = var * (n / max(n - 1, 1))
            updates[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            updates[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased
            updates[prefix + ".num_batches_tracked"] = state[prefix + ".num_batches_tracked"] + 1
    return y


def relu6(x: torch.Tensor) -> torch.Tensor:
    """nn.ReLU6 = hardtanh(0, 6): gradient passes strictly inside (0, 6) (asr.py:37)."""
    return F.relu6(x)


def inverted_residual(x: torch.Tensor, state: State, prefix: str, kernel_size: int,
                      stride: int = 1, use_residual: bool = True, training: bool = False,
                      updates: Optional[BNUpdates] = None) -> torch.Tensor:
    """x [B, Cin, T] -> [B, Cout, (T-1)//stride + 1]  (asr.py:40-59).

    pw (1x1, no bias) -> BN -> ReLU6 -> dw (groups=hid, pad (k-1)//2) -> BN -> ReLU6
    -> pw-linear (1x1, no bias) -> BN; plus x when use_residual.
    """
    p = prefix + ".conv"
    h = F.conv1d(x, state[p + ".0.0.weight"])
    h = relu6(_batch_norm(h, state, p + ".0.1", training, updates))
    hid = h.shape[1]
    h = F.conv1d(h, state[p + ".1.0.weight"], stride=stride,
                 padding=(kernel_size - 1) // 2, groups=hid)
    h = relu6(_batch_norm(h, state, p + ".1.1", training, updates))
    h = F.conv1d(h, state[p + ".2.weight"])
    h = _batch_norm(h, state, p + ".3", training, updates)
    return x + h if use_residual else h


def conv_voice_encoder(x: torch.Tensor, state: State, prefix: str, training: bool = False,
                       updates: Optional[BNUpdates] = None) -> torch.Tensor:
    """[B, Cin, T] -> [B, Cout, (T+1)//2]  (asr.py:62-79). Channel sizes are
    implied by the weights; kernels/strides/residual flags by the fixed table."""
    for i, k in enumerate(ENCODER_KERNELS):
        pre = f"{prefix}.layers.{i}"
        cin = state[pre + ".conv.0.0.weight"].shape[1]
        cout = state[pre + ".conv.2.weight"].shape[0]
        stride = 2 if i == 0 else 1
        residual = i in (1, 2, 3, 5, 6, 7)
        assert (cin == cout) or not residual
        x = inverted_residual(x, state, pre, k, stride, residual, training, updates)
    return x


def encoder_output_length(embed_len: torch.Tensor) -> torch.Tensor:
    """asr.py:81-82 -- integer, exact."""
    return torch.div(embed_len + 1, 2, rounding_mode="trunc")


def linear_char_decoder(x: torch.Tensor, state: State, prefix: str,
                        dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dropout(0.2) then Conv1d(k=1, bias)  (asr.py:85-94).  The dropout mask is
    injected (already scaled by 1/(1-p)) so train-mode parity is reproducible."""
    if dropout_mask is not None:
        x = x * dropout_mask
    return F.conv1d(x, state[prefix + ".layers.1.weight"], state[prefix + ".layers.1.bias"])


def audio_to_text_ctc_forward(audio: torch.Tensor, state: State, training: bool = False,
                              updates: Optional[BNUpdates] = None,
                              dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """audio [B, T, audio_size] f32 -> logits [B, (T+1)//2, vocab]  (asr.py:110-116)."""
    x = audio.transpose(1, 2)
    x = conv_voice_encoder(x, state, "encoder", training, updates)
    x = linear_char_decoder(x, state, "decoder", dropout_mask)
    return x.transpose(1, 2)


def ctc_loss_from_logits(logits: torch.Tensor, audio_len: torch.Tensor, text: torch.Tensor,
                         text_len: torch.Tensor) -> torch.Tensor:
    """asr.py:146-152: output_length, [T',B,V] log_softmax, CTCLoss(blank 0, mean,
    zero_infinity=True)."""
    logits_len = encoder_output_length(audio_len)
    log_probs = F.log_softmax(logits.transpose(0, 1), dim=-1)
    return F.ctc_loss(log_probs, text, logits_len, text_len, blank=0, reduction="mean",
                      zero_infinity=True)


def audio_to_text_ctc_loss(batch, state: State, training: bool = True,
                           updates: Optional[BNUpdates] = None,
                           dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """asr.py:133-152 without the augmentation call (tested separately with
    injected decisions: oracle.augment)."""
    (audio, audio_len), (text, text_len) = batch
    logits = audio_to_text_ctc_forward(audio, state, training, updates, dropout_mask)
    return ctc_loss_from_logits(logits, audio_len, text, text_len)


# --------------------------------------------------------------------- TTS --
def voice_decoder(x: torch.Tensor, state: State, prefix: str, training: bool = False,
                  updates: Optional[BNUpdates] = None) -> torch.Tensor:
    """[B, H, L] -> [B, out, 2L-1]  (tts.py:13-29): 4 IR @H, ConvTranspose1d(H->H/2,
    k5, s2, p2, bias), 3 IR @H/2, Conv1d(H/2->out, k1, bias)."""
    for i, k in enumerate(DECODER_KERNELS_A):
        x = inverted_residual(x, state, f"{prefix}.layers.{i}", k, 1, True, training, updates)
    x = F.conv_transpose1d(x, state[f"{prefix}.layers.4.weight"], state[f"{prefix}.layers.4.bias"],
                           stride=2, padding=2)
    for j, k in enumerate(DECODER_KERNELS_B):
        x = inverted_residual(x, state, f"{prefix}.layers.{5 + j}", k, 1, True, training, updates)
    return F.conv1d(x, state[f"{prefix}.layers.8.weight"], state[f"{prefix}.layers.8.bias"])


def align_text_to_audio_forward(aligntext: torch.Tensor, state: State, training: bool = False,
                                updates: Optional[BNUpdates] = None):
    """tts.py:172-190. Returns (hasf0_logits [B,T], f0_hat [B,T], logspc_hat [B,T,S],
    codeap_hat [B,T,1]) with S = 257 or 25 read off the norm stats."""
    x = F.embedding(aligntext, state["embedding.weight"]).transpose(1, 2)
    x = voice_decoder(x, state, "decoder", training, updates).transpose(1, 2)
    s = state["norm.logspc_mean"].shape[0]
    c = state["norm.codeap_mean"].shape[0]
    hasf0, f0, logspc, codeap = torch.split(x, [1, 1, s, c], dim=2)
    return hasf0[:, :, 0], f0[:, :, 0], logspc, codeap


def world_unnormalize(f0, logspc, codeap, state: State, prefix: str = "norm"):
    """_layers_v1.py:131-138."""
    return (state[prefix + ".f0_std"] * f0 + state[prefix + ".f0_mean"],
            state[prefix + ".logspc_std"] * logspc + state[prefix + ".logspc_mean"],
            state[prefix + ".codeap_std"] * codeap + state[prefix + ".codeap_mean"])


def world_normalize(f0, logspc, codeap, state: State, prefix: str = "norm"):
    """_layers_v1.py:122-129."""
    return ((f0 - state[prefix + ".f0_mean"]) / state[prefix + ".f0_std"],
            (logspc - state[prefix + ".logspc_mean"]) / state[prefix + ".logspc_std"],
            (codeap - state[prefix + ".codeap_mean"]) / state[prefix + ".codeap_std"])


def align_text_to_audio_predict(aligntext: torch.Tensor, state: State):
    """tts.py:192-201: unnormalise, then f0 := 0 where the has-f0 logit is < 0."""
    hasf0, f0, logspc, codeap = align_text_to_audio_forward(aligntext, state, False)
    f0, logspc, codeap = world_unnormalize(f0, logspc, codeap, state)
    f0 = torch.where(hasf0 < 0, torch.zeros((), dtype=f0.dtype), f0)
    return f0, logspc, codeap


def padding_mask(width: int, length: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """_layers_v1.py:14-24 / asr.py:15-24."""
    return (torch.arange(width)[None, :] < length[:, None]).to(dtype)


def mel_slope_weights(sample_rate: int = 16000, n_fft: int = 512) -> torch.Tensor:
    """_layers_v1.py:60-65."""
    f = (sample_rate / n_fft) * torch.arange(n_fft // 2 + 1, dtype=torch.float32)
    dm = 1127 / (700 + f)
    return dm / torch.sum(dm)


def world_loss(length, hasf0_logits, f0_hat, logspc_hat, codeap_hat, hasf0, f0, logspc, codeap,
               use_mel_weights: bool, loss: str = "mse"):
    """_layers_v1.py:69-93 (truncate to the common length, masked means)."""
    n = min(f0_hat.shape[1], f0.shape[1])
    hasf0_logits, f0_hat, logspc_hat, codeap_hat = (t[:, :n] for t in (hasf0_logits, f0_hat, logspc_hat, codeap_hat))
    hasf0, f0, logspc, codeap = (t[:, :n] for t in (hasf0, f0, logspc, codeap))
    if loss == "mse":
        el = lambda a, b: (a - b) ** 2
    elif loss == "l1":
        el = lambda a, b: (a - b).abs()
    else:
        raise ValueError("Unknown loss type")
    mask = padding_mask(n, length, f0.dtype)
    l_hasf0 = F.binary_cross_entropy_with_logits(hasf0_logits, hasf0, reduction="none") * mask
    l_f0 = el(f0_hat, f0) * hasf0 * mask
    if use_mel_weights:
        l_spc = torch.sum(el(logspc_hat, logspc) * mel_slope_weights()[None, None, :], dim=2) * mask
    else:
        l_spc = torch.mean(el(logspc_hat, logspc), dim=2) * mask
    l_ap = torch.mean(el(codeap_hat, codeap), dim=2) * mask
    ms = mask.sum()
    return l_hasf0.sum() / ms, l_f0.sum() / ms, l_spc.sum() / ms, l_ap.sum() / ms


def align_text_to_audio_loss(batch, state: State, use_mcep: bool, training: bool = True,
                             updates: Optional[BNUpdates] = None):
    """tts.py:203-213."""
    (f0, f0_len, logspc, codeap), (aligntext, aligntext_len) = batch
    hasf0 = (f0 >= 30.0).to(torch.float32)
    f0, logspc, codeap = world_normalize(f0, logspc, codeap, state)
    out = align_text_to_audio_forward(aligntext, state, training, updates)
    return world_loss(f0_len, *out, hasf0, f0, logspc, codeap, use_mel_weights=not use_mcep)


def text_to_align_text_forward(text: torch.Tensor, state: State, training: bool = False,
                               updates: Optional[BNUpdates] = None) -> torch.Tensor:
    """tts.py:79-87: Embedding -> 4 IR (k=5,11,17,29) -> Conv1d(H->2) ; [B,L] -> [B,L,2]."""
    x = F.embedding(text, state["embedding.weight"]).transpose(1, 2)
    for i, k in enumerate(ALIGN_KERNELS):
        x = inverted_residual(x, state, f"layers.{i}", k, 1, True, training, updates)
    x = F.conv1d(x, state["layers.4.weight"], state["layers.4.bias"])
    return x.transpose(1, 2)


def text_to_align_text_loss(batch, state: State, training: bool = True,
                            updates: Optional[BNUpdates] = None) -> torch.Tensor:
    """tts.py:121-130."""
    (text, text_len), (align, align_len) = batch
    align = align[:, :-1].reshape([align.shape[0], -1, 2])
    pred = text_to_align_text_forward(text, state, training, updates)
    logalign = torch.log((align + 1).to(pred.dtype))
    loss = torch.mean(torch.abs(logalign - pred), dim=2)
    mask = padding_mask(text.shape[1], text_len, pred.dtype)
    return torch.sum(loss * mask) / torch.sum(mask)


# --- v2 conv blocks ---------------------------------------------------
LN_EPS = 1e-5          # nn.LayerNorm default, _layers_v2.py:40


def conv_layer_block(x: torch.Tensor, state: State, prefix: str, transpose: bool, stride: int, padding: int) -> torch.Tensor:
    """ConvLayerBlock / ConvTransposeLayerBlock.forward (_layers_v2.py:50-56, 83-89): conv -> LayerNorm over the
    channel axis (via the two transposes) -> exact GELU.  x [B, Cin, T]."""
    w = state[f"{prefix}conv.weight"]
    b = state.get(f"{prefix}conv.bias")
    if transpose:
        y = F.conv_transpose1d(x, w, b, stride=stride, padding=padding)
    else:
        y = F.conv1d(x, w, b, stride=stride, padding=padding)
    c = y.shape[1]
    y = F.layer_norm(y.transpose(-2, -1), (c,), state[f"{prefix}layer_norm.weight"], state[f"{prefix}layer_norm.bias"], LN_EPS)
    return F.gelu(y.transpose(-2, -1))


def conv_layers(x: torch.Tensor, state: State, settings: Sequence[Sequence], prefix: str = "") -> torch.Tensor:
    """get_conv_layers(...) as an nn.Sequential (_layers_v2.py:92-106); settings rows are
    (out_channels, transpose, kernel_size, stride, padding, bias)."""
    for i, (_out, transpose, _k, stride, padding, _bias) in enumerate(settings):
        x = conv_layer_block(x, state, f"{prefix}{i}.", bool(transpose), int(stride), int(padding))
    return x
