#!/usr/bin/env python3
"""STAND-IN vectors with the layout of tests/golden/make_thirdparty_vectors.py's output, computed by THE ORACLE ITSELF (no pyworld, no
torchaudio): they pin nothing -- they exist only to execute tests/test_thirdparty_pins.py / test_gpu_thirdparty_pins.py end to end
(shapes, key names, calling conventions) in an image that has neither library.  Never written under tests/golden/.
usage: python tools/standin_thirdparty_vectors.py /tmp/standin && V100_THIRDPARTY_DIR=/tmp/standin python -m pytest tests/test_thirdparty_pins.py"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mel as omel, world_analysis as wa, world_synth as ws  # noqa: E402

out = sys.argv[1]
assert os.path.abspath(out) != os.path.join(ROOT, "tests", "golden"), "stand-ins never go under tests/golden/"
os.makedirs(out, exist_ok=True)
spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_thirdparty_vectors.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
sig = mk.synthetic()
sig = {k: v[: int(1.2 * 16000)] for k, v in sig.items()}       # short: the numpy restatement is slow
d = {"pyworld_version": np.array("STAND-IN (oracle)"), "names": np.array(sorted(sig))}
for n, x in sig.items():
    f0, tpos = wa.dio(x, 16000, f0_floor=80.0, f0_ceil=400.0, frame_period=10.0)        # (with the noise floor: as published the restatement
    # extrapolates F0 into exact silence, where pyworld has FFT rounding noise -- exactly the frames the pin tests exempt)
    sp = wa.cheaptrick(x, f0, tpos, 16000, fft_size=512)
    ap = wa.d4c(x, f0, tpos, 16000, fft_size=512)
    codeap = wa.code_aperiodicity(ap, 16000)
    dap = ws.decode_aperiodicity(codeap, 16000, 512)
    y = ws.synthesize(f0, sp, dap, 16000, frame_period=10.0)
    d.update({f"x_{n}": x, f"f0_{n}": f0, f"tpos_{n}": tpos, f"sp_{n}": sp, f"ap_{n}": ap, f"codeap_{n}": codeap, f"dap_{n}": dap, f"y_{n}": y})
np.savez_compressed(os.path.join(out, "thirdparty_world.npz"), **d)
m = {}
rng = np.random.RandomState(1)
for n, secs in (("1s", 1.0), ("10s", 10.0)):
    w = (0.3 * np.sin(2 * np.pi * 300.0 * np.arange(int(secs * 16000)) / 16000) + 0.05 * rng.randn(int(secs * 16000))).astype(np.float32)
    m[f"w_{n}"] = w
    m[f"logmel_{n}"] = omel.log_mel(w)
np.savez_compressed(os.path.join(out, "thirdparty_mel.npz"), **m)
print("stand-in vectors in", out)
