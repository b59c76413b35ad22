"""ctypes binding of libvoice100_hip.so (the C ABI in include/voice100_hip.h).

The library is built in-tree by `make` / `__graft_entry__.build()`; there is no
fallback: if it is missing or a kernel reports an error, a RuntimeError is raised.
Prototypes are read from the header itself so the binding cannot drift from it.
"""
import ctypes
import os
import re
import threading

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("VOICE100_LIB") or os.path.join(_PKG, "libvoice100_hip.so")     # VOICE100_LIB: an A/B build of the library
HEADER_PATH = os.path.join(_ROOT, "include", "voice100_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
    "double": ctypes.c_double,
}

_lock = threading.Lock()
_lib = None
_protos = None

STATUS = {1: "invalid or unsupported shape/mode", 2: "kernel launch error", 3: "required pointer is NULL"}


def parse_header(path=HEADER_PATH):
    """{name: [(ctype, argname), ...]} for every `int v100_*(...)` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|long long)\s+(v100_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2)
        params = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            if "*" in a:
                params.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, nm = a.rsplit(" ", 1)
                ty = ty.replace("const ", "").strip()
                params.append((_CTYPES[ty], nm))
        protos[name] = params
    return protos


def load():
    """Load the shared library once; RuntimeError if it has not been built."""
    global _lib, _protos
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `make` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "voice100_amd has no CPU/PyTorch fallback for its kernels.")
        lib = ctypes.CDLL(LIB_PATH)
        protos = parse_header()
        for name, params in protos.items():
            fn = getattr(lib, name)        # AttributeError if the header declares a symbol the .so lacks
            fn.argtypes = [t for t, _ in params]
            fn.restype = ctypes.c_longlong if (name.endswith("_bytes") or name in ("v100_launch_count", "v100_ir_stack_plan", "v100_world_randn_bound")) else ctypes.c_int
        _lib, _protos = lib, protos
        return lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("voice100_amd kernels take CUDA (ROCm) tensors only")
        if not x.is_contiguous():
            raise RuntimeError("voice100_amd kernels take contiguous tensors")
        p = x.data_ptr()
        if p % 16:
            raise RuntimeError("tensor storage must be 16-byte aligned")
        return p
    return x


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """torch's CURRENT stream on the current device as a raw hipStream_t (an int).  The private accessor is one C call; the
    public torch.cuda.current_stream().cuda_stream builds a Stream object per call (10 us x ~10 library calls a step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_fast = {}


def call(name, *args):
    """Launch `name` on torch's current stream (the trailing `stream` argument is appended here).
    Tensors must be contiguous CUDA tensors (checked); None is a NULL pointer."""
    ent = _fast.get(name)
    if ent is None:
        lib = load()
        params = _protos[name]
        ent = _fast[name] = (getattr(lib, name), [t is ctypes.c_void_p for t, _ in params],
                             bool(params) and params[-1][1] == "stream", len(params))
    fn, is_ptr, wants_stream, nparams = ent
    if wants_stream and len(args) == nparams - 1:
        args = args + (stream_ptr(),)
    if len(args) != nparams:
        raise TypeError(f"{name}: expected {nparams} arguments, got {len(args)}")
    conv = [(_ptr(a) if (p and a is not None and not isinstance(a, int)) else a) for a, p in zip(args, is_ptr)]
    rc = fn(*conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {STATUS.get(rc, rc)} (status {rc})")


def helper(name, *args):
    """Host-side helpers that return a count rather than a status (v100_*_num_* / _splits)."""
    lib = load()
    return getattr(lib, name)(*args)


TIMING_TAGS = {"dw_fwd": 0, "dw_bwd_data": 1, "dw_wgrad": 2, "pw_gemm": 3, "pw_wgrad": 4, "other": 5}


def launch_count() -> int:
    """Kernel launches the library has issued in this process so far."""
    return int(load().v100_launch_count())


def timing_enable(tags=True) -> None:
    """Opt-in HIP-event timing inside the library (see include/voice100_hip.h); bench.py only.
    tags: True = every tag, False/None = off, or an iterable of TIMING_TAGS names (each timed launch costs ~3 us)."""
    if tags is True:
        mask = (1 << len(TIMING_TAGS)) - 1
    elif not tags:
        mask = 0
    else:
        mask = 0
        for t in tags:
            mask |= 1 << TIMING_TAGS[t]
    load().v100_timing_enable(mask)


def timing_read():
    """{tag: (launches, total_ms, algorithmic_bytes)} recorded since timing_enable(True)."""
    lib = load()
    out = {}
    for name, tag in TIMING_TAGS.items():
        ms, n, nb = ctypes.c_double(0.0), ctypes.c_longlong(0), ctypes.c_double(0.0)
        lib.v100_timing_read(tag, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(nb))
        if n.value:
            out[name] = (n.value, ms.value, nb.value)
    return out
