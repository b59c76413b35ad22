"""Log-mel batch augmentation with the random decisions injected.
TEST INFRASTRUCTURE.  Reference: voice100/audio.py:17-108.

Each function takes the values the reference draws from `random` / `torch.rand`
as arguments, so the HIP path and this restatement see the same decisions.
"""
import math
import torch

LOG_OFFSET = 1e-6                       # audio.py:14
BLANK_AUDIO = math.log(LOG_OFFSET)      # audio.py:25


def timestretch(audio, audio_len, rate: int):
    """audio.py:52-58; rate in [50,150): new T = T*rate//100, nearest-lower frame."""
    new_t = audio.shape[1] * rate // 100
    new_len = torch.div(audio_len * rate, 100, rounding_mode="trunc")
    idx = torch.div(torch.arange(new_t) * 100, rate, rounding_mode="trunc")
    return audio[:, idx, :], new_len


def pitchshift(audio, rate: float):
    """audio.py:60-64: mel bin i reads bin int(i*rate) clamped."""
    idx = (torch.arange(audio.shape[2]) * rate).int().clamp(0, audio.shape[2] - 1).long()
    return audio[:, :, idx]


def ampshift(audio, rate: float):
    """audio.py:66-68."""
    return audio - rate


def timemask(audio, spans):
    """audio.py:70-80; spans = [(t, hw, a)], python slice semantics (negative start
    counts from the end)."""
    audio = audio.clone()
    for t, hw, a in spans:
        audio[:, int(t - hw):int(t + hw), :] = a
    return audio


def freqmask(audio, t: int, hw: int, a: float):
    """audio.py:82-90."""
    audio = audio.clone()
    audio[:, :, int(t - hw):int(t + hw)] = a
    return audio


def mixnoise(audio, low: float, high: float, std: float, uniform):
    """audio.py:92-98; `uniform` is the torch.rand(audio.shape) draw."""
    scale = torch.linspace(low, high, 64)[None, :]
    noise = uniform * std + scale
    return torch.log(torch.clamp(torch.exp(audio) + torch.exp(noise), min=LOG_OFFSET))


def _time_mask(audio, audio_len):
    return (torch.arange(audio.shape[1])[None, :, None] < audio_len[:, None, None]).float()


def mixaudio(audio, audio_len):
    """audio.py:100-104: 0.9*x + 0.1*(next utterance in the batch), masked, in the
    linear domain."""
    m = _time_mask(audio, audio_len)
    x = torch.exp(audio) * m
    y = torch.roll(x, shifts=-1, dims=0)
    return torch.log(torch.clamp((0.9 * x + 0.1 * y) * m, min=LOG_OFFSET))


def maskaudio(audio, audio_len):
    """audio.py:106-108: padding -> log(1e-6); real frames exp/log round trip."""
    m = _time_mask(audio, audio_len)
    return torch.log(torch.clamp(torch.exp(audio) * m, min=LOG_OFFSET))
