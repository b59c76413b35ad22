"""Integer / index restatements (bit-exact targets).  TEST INFRASTRUCTURE.

Reference anchors:
  output_length            voice100/models/asr.py:81-82, 118-122
  generate_padding_mask    voice100/models/_layers_v1.py:14-24
  TextToAlignTextModel.align   voice100/models/tts.py:89-110
  ctc_best_path            voice100/models/align.py:18-66
  CharTokenizer.merge_repeated voice100/text.py:99-104
"""
import numpy as np


def output_length(audio_len):
    """(len + 1) // 2 with truncation (asr.py:82); lengths are non-negative."""
    a = np.asarray(audio_len)
    return (a + 1) // 2


def padding_mask(width, length):
    """1.0 where position < length (_layers_v1.py:24)."""
    length = np.asarray(length)
    return (np.arange(width)[None, :] < length[:, None]).astype(np.float32)


def expand_align(text, align, head=5, tail=5):
    """tts.py:89-110: lay each token text[i] over [round(t+gap), round(t+gap+len)),
    at least one frame, `head`/`tail` blank frames around.  Python's round()
    (banker's rounding on the float accumulator) is part of the behaviour."""
    text = np.asarray(text)
    align = np.asarray(align)
    assert text.ndim == 1 and align.ndim == 2
    total = head + int(np.sum(align)) + tail
    out = np.zeros(total, dtype=text.dtype)
    t = head
    for i in range(align.shape[0]):
        t += align[i, 0].item()
        s = round(t)
        t += align[i, 1].item()
        e = round(t)
        if s == e:
            e = max(0, e + 1)
        for j in range(s, e):
            out[j] = text[i]
    return out


def ctc_best_path(logits, labels, max_move=3):
    """Forced alignment by a banded Viterbi over the blank-expanded label row
    (align.py:18-66).  At most `max_move-1` label positions are skipped per frame,
    a 2-step move may not land on a blank, ties resolve to the smallest move.
    Returns (best_score, best_path[int32 T], best_labels[T])."""
    logits = np.asarray(logits)
    labels = np.asarray(labels)
    T = logits.shape[0]
    ext = np.zeros(2 * labels.shape[0] + 1, dtype=labels.dtype)
    ext[1::2] = labels
    n = ext.shape[0]

    back = [np.full(2, -1, dtype=np.int32)]
    score = np.array([logits[0, ext[0]], logits[0, ext[1]]], dtype=logits.dtype)
    for i in range(1, T):
        width = min(score.shape[0] + max_move - 1, n)
        nscore = np.full(width, -np.inf, dtype=score.dtype)
        nback = np.zeros(width, dtype=np.int32)
        for v in range(width):
            best, arg = -np.inf, 0
            first = True
            for j in range(max_move):
                k = v - j
                cand = -np.inf
                if 0 <= k < score.shape[0] and v < n:
                    cand = score[k] + logits[i, ext[v]]
                    if j > 0 and j % 2 == 0 and ext[v] == 0:
                        cand = -np.inf
                    src = k
                else:
                    src = 0
                if first or cand > best:
                    best, arg, first = cand, src, False
            nscore[v] = best
            nback[v] = arg
        score = nscore
        back.append(nback)
    path = np.zeros(T, dtype=np.int32)
    j = n + (-1 if score[-1] > score[-2] else -2)
    best_score = score[j]
    for i in range(T - 1, -1, -1):
        path[i] = j
        j = back[i][j]
    return best_score, path, ext[path]


def merge_repeated_ids(ids, blank=0):
    """CTC collapse on token ids: drop repeats, then blanks (text.py:99-104 does the
    same on the decoded string with a regex; ' ' alone -> '' is a string-level quirk
    handled by the caller)."""
    out, prev = [], None
    for x in list(ids):
        if x != prev and x != blank:
            out.append(int(x))
        prev = x
    return out
