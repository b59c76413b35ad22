#!/usr/bin/env python3
"""tests/golden/world_ref_samples.npz from the reference's DATA files docs/sample-*.wav (run where /root/reference exists).

Those WAVs are outputs of the reference's TTS chain -- waveforms pyworld.synthesize produced from model-predicted WORLD features (16 kHz, mono,
int16, 10 ms frames) -- and the only artefacts of pyworld's arithmetic in the build image.  Stored: one whole utterance (en1) and the first
1.5 s of two that begin unvoiced (ja1_head, en2_head), as int16 sample arrays.  What they pin: tests/golden/README.md."""
import os
import sys
import wave

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def read(name, n=None):
    w = wave.open(os.path.join(REF, "docs", name))
    assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
    return np.frombuffer(w.readframes(w.getnframes() if n is None else n), dtype=np.int16).copy()


np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "world_ref_samples.npz"),
                    en1=read("sample-en-1.wav"), ja1_head=read("sample-ja-1.wav", 24000), en2_head=read("sample-en-2.wav", 24000),
                    sample_rate=np.int32(16000))
