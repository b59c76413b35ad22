"""Integer decode / alignment steps on the GPU, bit-exact with the reference:
greedy CTC decoding (argmax + merge_repeated, voice100/text.py:99-104), ctc_best_path forced alignment
(voice100/models/align.py:18-66) and TextToAlignTextModel.align (voice100/models/tts.py:89-110)."""
import torch

from . import _native as N


def ctc_greedy_decode(logits: torch.Tensor, lengths: torch.Tensor = None, blank: int = 0):
    """logits [B, T, V] fp32 -> (ids [B, T] int64 zero-padded, lens [B] int32)."""
    if not logits.is_cuda:
        raise RuntimeError("ctc_greedy_decode: GPU tensors only")
    logits = logits.contiguous().float()
    B, T, V = logits.shape
    lens = lengths.to(device=logits.device, dtype=torch.int32).contiguous() if lengths is not None else None
    out = torch.empty((B, T), dtype=torch.int64, device=logits.device)
    out_len = torch.empty((B,), dtype=torch.int32, device=logits.device)
    N.call("v100_ctc_greedy_decode", logits, lens, out, out_len, B, T, V, int(blank))
    return out, out_len


def ctc_best_path(log_probs: torch.Tensor, labels: torch.Tensor, input_lengths=None, label_lengths=None, max_move: int = 3):
    """log_probs [B, T, V] fp32, labels [B, L] int64 -> (score [B], path [B, T] int32, best_labels [B, T] int64)."""
    if not log_probs.is_cuda:
        raise RuntimeError("ctc_best_path: GPU tensors only")
    lp = log_probs.contiguous().float()
    B, T, V = lp.shape
    labels = labels.to(device=lp.device, dtype=torch.int64).contiguous()
    L = labels.shape[1]
    il = input_lengths.to(device=lp.device, dtype=torch.int32).contiguous() if input_lengths is not None else None
    ll = label_lengths.to(device=lp.device, dtype=torch.int32).contiguous() if label_lengths is not None else None
    back = torch.empty((B, T, 2 * L + 1), dtype=torch.int16, device=lp.device)
    path = torch.empty((B, T), dtype=torch.int32, device=lp.device)
    score = torch.empty((B,), dtype=torch.float32, device=lp.device)
    N.call("v100_ctc_best_path", lp, labels, il, ll, back, path, score, B, T, V, L, int(max_move))
    ext = torch.zeros((B, 2 * L + 1), dtype=torch.int64, device=lp.device)
    ext[:, 1::2] = labels
    return score, path, torch.gather(ext, 1, path.long())


def align_expand(text: torch.Tensor, align: torch.Tensor, text_len=None, head: int = 5, tail: int = 5):
    """text [B, L] int64, align [B, L, 2] (gap, length) -> (aligntext [B, Tmax] int64 zero-padded, lens [B] int32)."""
    if not text.is_cuda:
        raise RuntimeError("align_expand: GPU tensors only")
    text = text.to(torch.int64).contiguous()
    al = align.to(device=text.device, dtype=torch.float64).contiguous()
    B, L = text.shape
    tl = text_len.to(device=text.device, dtype=torch.int32).contiguous() if text_len is not None else None
    out_len = torch.empty((B,), dtype=torch.int32, device=text.device)
    # pass 1: the lengths only (the output's width depends on the data: ONE read-back), pass 2: the expansion, exactly that wide
    # (pad_sequence semantics: the longest utterance ends at the tensor's edge)
    N.call("v100_align_expand", text, al, tl, None, out_len, B, L, 0, int(head), int(tail))
    tmax = max(int(out_len.max()), 1)
    out = torch.empty((B, tmax), dtype=torch.int64, device=text.device)
    N.call("v100_align_expand", text, al, tl, out, out_len, B, L, tmax, int(head), int(tail))
    return out, out_len
