"""Integer outputs on the device, bit-exact: greedy CTC decode, ctc_best_path (golden vectors from the reference's
numpy implementation) and align() expansion (golden + oracle on random inputs)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import intops

pytestmark = pytest.mark.gpu


def test_ctc_best_path_golden_and_random(cuda):
    from voice100_amd.decode import ctc_best_path
    g = load_golden("int_tables.npz")
    for n in range(3):
        lp = torch.from_numpy(g[f"ctc{n}/logits"])[None].to(cuda)
        lab = torch.from_numpy(g[f"ctc{n}/labels"])[None].to(cuda)
        score, path, best = ctc_best_path(lp, lab)
        assert np.array_equal(path[0].cpu().numpy(), g[f"ctc{n}/path"])                  # bit-exact
        assert np.array_equal(best[0].cpu().numpy(), g[f"ctc{n}/best_labels"])
        assert float(score[0]) == pytest.approx(float(g[f"ctc{n}/score"]), rel=1e-6)
    rng = np.random.RandomState(5)
    B, T, V, L = 6, 90, 29, 17
    lp = torch.log_softmax(torch.from_numpy(rng.randn(B, T, V).astype(np.float32)), -1)
    labels = torch.from_numpy(rng.randint(1, V, size=(B, L)))
    lens = torch.tensor([90, 77, 60, 90, 45, 88], dtype=torch.int32)
    llen = torch.tensor([17, 9, 17, 1, 12, 5], dtype=torch.int32)
    score, path, best = ctc_best_path(lp.to(cuda), labels.to(cuda), lens, llen)
    for b in range(B):
        s, p, bl = intops.ctc_best_path(lp[b, :lens[b]].numpy(), labels[b, :llen[b]].numpy())
        assert np.array_equal(path[b, :lens[b]].cpu().numpy(), p), b
        assert np.array_equal(best[b, :lens[b]].cpu().numpy(), bl), b


def test_greedy_decode(cuda):
    from voice100_amd.decode import ctc_greedy_decode
    g = torch.Generator().manual_seed(2)
    B, T, V = 5, 700, 29
    logits = torch.randn(B, T, V, generator=g)
    logits[:, :, 0] += 1.5                                   # plenty of blanks
    logits[0, 10:40] = logits[0, 10:11]                      # a long run of one symbol
    lens = torch.tensor([700, 512, 1, 257, 699], dtype=torch.int32)
    ids, n = ctc_greedy_decode(logits.to(cuda), lens)
    for b in range(B):
        ref = intops.merge_repeated_ids(logits[b, :lens[b]].argmax(-1).tolist())
        assert int(n[b]) == len(ref)
        assert ids[b, :len(ref)].cpu().tolist() == ref
        assert int(ids[b, len(ref):].abs().sum()) == 0
    # token ids agree with the golden argmax of the ASR fixture
    a = load_golden("asr_tiny.npz")
    ids, n = ctc_greedy_decode(torch.from_numpy(a["logits_eval"]).to(cuda))
    for b in range(a["argmax_eval"].shape[0]):
        assert ids[b, :int(n[b])].cpu().tolist() == intops.merge_repeated_ids(a["argmax_eval"][b].tolist())


def test_align_expand(cuda):
    from voice100_amd.decode import align_expand
    a = load_golden("align_tiny.npz")
    for k in range(3):
        text = torch.from_numpy(a[f"align_case{k}/text"])[None].to(cuda)
        al = torch.from_numpy(a[f"align_case{k}/align"]).double()[None].to(cuda)
        out, n = align_expand(text, al)
        ref = a[f"align_case{k}/aligntext"]
        assert int(n[0]) == len(ref) and np.array_equal(out[0, :len(ref)].cpu().numpy(), ref)
    rng = np.random.RandomState(3)
    B, L = 7, 40
    text = torch.from_numpy(rng.randint(1, 29, size=(B, L)))
    al = torch.from_numpy(np.round(rng.rand(B, L, 2) * 4, 1))           # .5 ties on purpose
    tl = torch.tensor([40, 33, 1, 40, 17, 25, 8], dtype=torch.int32)
    out, n = align_expand(text.to(cuda), al.to(cuda), tl)
    for b in range(B):
        ref = intops.expand_align(text[b, :tl[b]].numpy(), al[b, :tl[b]].numpy())
        assert int(n[b]) == len(ref), b
        assert np.array_equal(out[b, :len(ref)].cpu().numpy(), ref), b
