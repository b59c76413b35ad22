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


def test_stack_plan_layout(lib):
    """v100_ir_stack_plan (host arithmetic only): the encoder's nine blocks at the benchmark shape and at a time-stretched odd length --
    every tensor 256-byte aligned inside the blob, no two overlapping, sized for its storage format, and the totals consistent with
    the per-block helpers."""
    from voice100_amd import functional as F_
    cfgs = ((64, 256, 256, 11, 2, 0), (256, 1024, 256, 19, 1, 1), (256, 1024, 256, 27, 1, 1), (256, 1024, 256, 35, 1, 1),
            (256, 1024, 512, 51, 1, 0), (512, 2048, 512, 59, 1, 1), (512, 2048, 512, 67, 1, 1), (512, 2048, 512, 75, 1, 1),
            (512, 2048, 512, 83, 1, 0))
    for T in (1024, 759):
        for bf16, level in ((0, 0), (1, 0), (1, 2), (1, 4)):
            desc, blocks, totals = F_._stack_plan(cfgs, 32, T, bf16, level, False)
            B, t = 32, T
            spans = []
            for (cin, hid, cout, k, s, res), o in zip(cfgs, blocks):
                t2 = (t + 2 * ((k - 1) // 2) - k) // s + 1
                assert o[7] == t2
                act = bf16 == 1 and level and s == 1
                P, P2 = F_.pitch16(t, B), F_.pitch16(t2, B)
                sizes = {0: B * hid * (P * 2 if act else t * 4), 1: B * hid * (P * 2 if act else t2 * 4),
                         2: B * cout * (P * 2 if (act and level >= 3) else t2 * 4), 3: B * cout * t2 * 4,
                         5: 12 * max(hid, cout) * 4}
                if o[4] >= 0:
                    sizes[4] = B * cout * P2 * 2
                for j, n in sizes.items():
                    assert o[j] % 256 == 0 and o[j] >= 0
                    spans.append((o[j], o[j] + n))
                spans.append((o[6], o[6] + 1))
                assert (o[4] >= 0) == (bf16 == 1 and level >= 4 and (cin, k) != (512, 83))      # every block but the last writes a shadow
                t = t2
            spans.sort()
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 <= b0, (T, bf16, level, a0, a1, b0, b1)
            assert totals[0] >= spans[-1][1] and totals[1] > 0
            assert totals[2] == sum(hid * cin + 2 * hid + hid * k + 2 * hid + cout * hid + 2 * cout for cin, hid, cout, k, _, _ in cfgs)
    bad = (ctypes.c_int * 12)(1, 32, 0, 1, 4, 0, 64, 256, 256, 11, 2, 0)
    assert lib.v100_ir_stack_plan(bad, (ctypes.c_longlong * 14)()) < 0


def test_world_analysis_host_side(lib):
    """The host-only pieces of the WORLD analysis entry points (csrc/world_analysis.hip): frame count, DIO band plan, the randn
    sequence in double against the oracle's, table bounds, workspace sizes, and argument validation (all of which return before
    anything touches a device)."""
    import ctypes
    import numpy as np
    from oracle import world_analysis as wa
    from oracle import world_synth as ws
    for n in (1, 159, 160, 16000, 163680):
        assert lib.v100_world_frames(16000, n, 10.0) == wa.samples_for_dio(16000, n, 10.0)
    assert lib.v100_world_frames(22050, 22050, 5.0) == wa.samples_for_dio(22050, 22050, 5.0)
    assert lib.v100_world_frames(0, 10, 10.0) == -1
    half = (ctypes.c_int * 16)()
    nb = lib.v100_world_dio_bands(16000, 80.0, 400.0, 2.0, half, 16)
    assert nb == 5 == 1 + int(np.log(400.0 / 80.0) / wa.K_LOG2 * 2.0)
    assert [half[i] for i in range(nb)] == [wa.matlab_round(16000 / (80.0 * 2.0 ** ((i + 1) / 2.0)) / 2.0) for i in range(nb)]
    assert lib.v100_world_dio_bands(16000, 71.0, 800.0, 2.0, half, 16) == 7
    assert lib.v100_world_dio_bands(16000, 400.0, 80.0, 2.0, half, 16) == -1          # ceil below floor
    assert lib.v100_world_dio_bands(16000, 71.0, 800.0, 2.0, half, 3) == -1           # more bands than the caller has room for
    tab = np.empty(4096, np.float64)
    assert lib.v100_world_randn_host_f64(tab.ctypes.data_as(ctypes.c_void_p), 4096) == 0
    assert np.array_equal(tab, np.asarray(ws.randn_table(4096), np.float64))
    assert abs(tab.mean()) < 0.1 and abs(tab.std() - 1.0) < 0.05
    assert lib.v100_world_randn_host_f64(None, 4) == 3
    # cheaptrick consumes at most (fft_size - 2) + fft_size/2 + 1 draws per frame, d4c a 3-period window at 40 Hz + three 4-period windows at 47 Hz
    assert lib.v100_world_randn_bound(0, 100, 16000, 512) >= 100 * (510 + 257)
    assert lib.v100_world_randn_bound(1, 100, 16000, 512) == 100 * ((2 * wa.matlab_round(1.5 * 16000 / 40.0) + 1) + 3 * (2 * wa.matlab_round(2.0 * 16000 / 47.0) + 1))
    assert lib.v100_world_dio_workspace_bytes(16, 163680, 16000, 80.0, 400.0, 2.0, 10.0) > 16 * 163680 * 8 * 6
    assert lib.v100_world_dio_workspace_bytes(0, 163680, 16000, 80.0, 400.0, 2.0, 10.0) == -1
    assert lib.v100_world_d4c_workspace_bytes(16, 163680, 16000, 10.0) >= 3 * 16 * 1024 * 8
    one = ctypes.c_void_p(16)
    assert lib.v100_world_dio(None, one, 1, 100, 100, 16000, 80.0, 400.0, 2.0, 10.0, 0.1, one, one, one, one, None) == 3
    assert lib.v100_world_dio(one, one, 1, 100, 50, 16000, 80.0, 400.0, 2.0, 10.0, 0.1, one, one, one, one, None) == 1       # pitch < max_len
    assert lib.v100_world_cheaptrick(one, one, one, 1, 100, 100, 16000, 10.0, -0.15, 768, one, 1000, one, one, None, 1e-15, one, None) == 1
    assert lib.v100_world_cheaptrick(one, one, one, 1, 100, 100, 16000, 10.0, -0.15, 512, one, 1000, one, None, None, 1e-15, one, None) == 3
    assert lib.v100_world_d4c(one, one, one, 1, 100, 100, 16000, 10.0, 0.85, 512, one, 1000, one, one, 700, one, None, None, one, None) == 1   # window length
    assert lib.v100_world_d4c(one, one, one, 1, 100, 100, 8000, 10.0, 0.85, 512, one, 1000, one, one, 769, one, None, None, one, None) == 1    # 8 kHz: no band
