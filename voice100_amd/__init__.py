"""voice100_amd -- MI355X-native implementation of the Voice100 CNN hot path.

Host side: PyTorch-ROCm modules with the reference's class names, constructor
arguments, forward() layouts and state_dict keys (voice100/models/asr.py, tts.py).
Device side: hand-written HIP kernels for gfx950 behind the C ABI declared in
include/voice100_hip.h (libvoice100_hip.so, loaded with ctypes).
"""
__version__ = "0.1.0"
