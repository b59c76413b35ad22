"""CPU-side checks of the boundary: the shared library loads and exports every symbol the header declares (no compute
calls -- there is no GPU here), the ctypes prototypes are derived from the header, the host-side helpers agree with the
oracle's integer arithmetic, and the product modules refuse to run without a GPU (no silent fallback)."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from voice100_amd import _native as N
    if not os.path.exists(N.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return N.load()


def test_every_declared_symbol_is_exported(lib):
    from voice100_amd import _native as N
    text = re.sub(r"/\*.*?\*/", "", open(N.HEADER_PATH).read(), flags=re.S)
    declared = re.findall(r"\b(v100_\w+)\s*\(", text)
    assert len(declared) >= 30 and len(set(declared)) == len(declared)
    for name in declared:
        assert hasattr(lib, name), name
    assert set(N.parse_header()) == set(declared)


def test_host_helpers_need_no_gpu(lib):
    assert lib.v100_pw_num_parts(32, 1024) == 32 * 8
    assert lib.v100_dw_num_groups(32, 2048) == 1 and lib.v100_dw_num_groups(32, 256) == 8 and lib.v100_dw_num_groups(2, 4) == 2
    assert 1 <= lib.v100_pw_wgrad_splits(32, 2048, 512) <= 32
    shape = (ctypes.c_int * 9)(32, 512, 2048, 512, 512, 83, 1, 1, 1)
    assert lib.v100_ir_prep_bytes(shape) >= 4 * 2048 * 512 * 2
    assert lib.v100_ir_fwd_workspace_bytes(shape) > 0 and lib.v100_ir_bwd_workspace_bytes(shape) > 2 * 32 * 2048 * 512 * 4
    # NULL pointers are reported, not dereferenced
    assert lib.v100_dwconv(None, None, None, None, None, None, 0, None, None, None, None, 3, None, 1, 1, 1, 8, 8, 3, 1, 1, 0, 1, 0, None) == 3
    assert lib.v100_pw_gemm(None, None, None, None, None, None, None, 0, None, None, None, None, None, 0, None, 1, 1, 1, 1, 0, None) == 3


def test_no_cpu_fallback():
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel
    from voice100_amd.mel import MelSpectrogramAudioTransform
    from voice100_amd.audio import BatchSpectrogramAugumentation
    with pytest.raises(RuntimeError):
        AudioToTextCTC(64, 32, 29, 32)(torch.zeros(1, 16, 64))
    with pytest.raises(RuntimeError):
        AlignTextToAudioModel(29, 32).predict(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(RuntimeError):
        MelSpectrogramAudioTransform()(torch.zeros(1600))
    with pytest.raises(RuntimeError):
        BatchSpectrogramAugumentation()(torch.zeros(1, 8, 64), torch.tensor([8]))


def test_state_dict_keys_match_reference_layout():
    from conftest import load_golden, sub
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import AlignTextToAudioModel, TextToAlignTextModel
    g = load_golden("asr_tiny.npz")
    m = AudioToTextCTC(64, 32, 29, 32)
    assert set(m.state_dict()) == set(sub(g, "state/"))
    m.load_state_dict(sub(g, "state/"), strict=True)
    assert sum(p.numel() for p in AudioToTextCTC(64, 512, 29, 512).parameters()) == 11621661        # README.md:135-147
    assert sum(p.numel() for p in AlignTextToAudioModel(29, 512).parameters()) == 11060234            # README.md:73-85
    assert sum(p.numel() for p in TextToAlignTextModel(29, 512).parameters()) == 8568322              # README.md:59-69
    g = load_golden("tts_tiny_mcep.npz")
    AlignTextToAudioModel(29, 32, use_mcep=True).load_state_dict(sub(g, "state/"), strict=True)


def test_output_length_and_align_are_integer_exact():
    from conftest import load_golden
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.tts import TextToAlignTextModel
    from oracle import intops
    g = load_golden("int_tables.npz")
    m = AudioToTextCTC(64, 32, 29, 32)
    assert np.array_equal(m.output_length(torch.from_numpy(g["output_length/in"])).numpy(), g["output_length/out"])
    a = load_golden("align_tiny.npz")
    t = TextToAlignTextModel(29, 32)
    for n in range(3):
        text, align = a[f"align_case{n}/text"], a[f"align_case{n}/align"]
        got = t.align(torch.from_numpy(text), torch.from_numpy(align)).numpy()
        assert np.array_equal(got, a[f"align_case{n}/aligntext"]) and np.array_equal(got, intops.expand_align(text, align))


def test_augment_draw_consumes_rng_like_the_reference():
    """Replay the reference's random-call order (audio.py:34-49) by hand against draw()."""
    from voice100_amd.audio import BatchSpectrogramAugumentation
    aug = BatchSpectrogramAugumentation()
    audio = torch.zeros(2, 40, 64)
    for seed in range(60):
        random.seed(seed)
        d = aug.draw(audio)
        after = random.random()
        random.seed(seed)
        T = 40
        if random.random() < 0.2:
            r = random.randrange(50, 150); assert d.stretch_rate == r; T = T * r // 100
        else:
            assert d.stretch_rate == 0
        if random.random() < 0.2:
            assert d.pitch_rate == 1.0 + random.random() * 0.2
        if random.random() < 0.2:
            assert d.amp == 1.0 + random.random() * 3.0
        if random.random() < 0.2:
            for i in range(random.randint(1, 3)):
                assert d.tmask[i] == (random.randrange(0, T), random.randint(1, 3), random.uniform(-aug.blank_audio, -5))
        if random.random() < 0.2:
            assert d.fmask == (random.randrange(0, 64), random.randint(1, 10), random.uniform(-aug.blank_audio, -5))
        if random.random() < 0.2:
            assert d.noise[:3] == (-5.0 + 5.0 * random.random(), -5.0 + 5.0 * random.random(), 5.0 * random.random())
        assert d.mix == (random.random() < 0.2)
        assert after == random.random()                       # same number of draws consumed
