#!/usr/bin/env python3
"""Timeline of the chained WORLD time base (csrc/world.hip, world_timebase_chain_kernel) for utterance 0 of a 16 x 10 s batch: per chunk
the wall-clock stamps at start / increments done / running phase received / phase handed on / wrap done / end.  Needs a library built with
-DTBC_DEBUG (compile csrc/world.hip with it, link against build/obj/*.o, point VOICE100_LIB at the result)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from voice100_amd.vocoder import WORLDVocoder
from voice100_amd import _native as N
dev = torch.device("cuda"); v = WORLDVocoder()
rng = np.random.RandomState(5); B, T = 16, 1023
f0 = np.where(np.sin(np.arange(T)[None] / 40.0 + rng.rand(B, 1) * 6) > 0.2, 0.0, 90 + 160 * rng.rand(B, 1) + 20 * np.sin(np.arange(T)[None] / 7.0)).astype(np.float32)
k = np.arange(257)
sp = (1e-2 * (1 + 4 * np.exp(-((k * 16000 / 512 - 1500) / 300.0) ** 2))[None, None] * np.exp(0.3 * rng.randn(B, T, 1))).astype(np.float32)
cod = np.where(f0[..., None] > 0, -10 - 25 * rng.rand(B, T, 1), 0.0).astype(np.float32)
f0, sp, cod = (torch.from_numpy(a).to(dev) for a in (f0, sp, cod))
for _ in range(3):
    y, n = v.synthesize(f0, sp, codeap=cod)
torch.cuda.synchronize()
buf = np.zeros(8 * 64, np.int64)
lib = N.load(); lib.v100_tbc_debug_read.argtypes = [ctypes.c_void_p]
assert lib.v100_tbc_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
d = buf.reshape(64, 8)[:21, :6].astype(np.float64)
t0 = d[:20, 0].min()
print("chunk: start, interp done, carry got, scan done(publish), fmod done, end   [us, 100 MHz clock]")
for c in range(21):
    print(c, np.round((d[c] - t0) / 100.0, 1).tolist())
