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
LIB_PATH = os.path.join(_PKG, "libvoice100_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "voice100_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
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
    for m in re.finditer(r"\bint\s+(v100_\w+)\s*\(([^)]*)\)\s*;", text):
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
            fn.restype = ctypes.c_int
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


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Launch `name` on torch's current stream (the trailing `stream` argument is appended here)."""
    lib = load()
    fn = getattr(lib, name)
    params = _protos[name]
    full = list(args)
    if params and params[-1][1] == "stream" and len(full) == len(params) - 1:
        full.append(stream_ptr())
    if len(full) != len(params):
        raise TypeError(f"{name}: expected {len(params)} arguments, got {len(full)}")
    conv = [(_ptr(a) if t is ctypes.c_void_p else a) for a, (t, _) in zip(full, params)]
    rc = fn(*conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {STATUS.get(rc, rc)} (status {rc})")


def helper(name, *args):
    """Host-side helpers that return a count rather than a status (v100_*_num_* / _splits)."""
    lib = load()
    return getattr(lib, name)(*args)


class KernelTimer:
    """Opt-in HIP-event timing of kernel regions on torch's current stream (bench.py only).
    Usage: N.timer = KernelTimer(); ... ; N.timer.summary() after a synchronize."""

    def __init__(self):
        self.events = {}

    def record(self, tag, start, end):
        self.events.setdefault(tag, []).append((start, end))

    def summary(self):
        torch.cuda.synchronize()
        return {tag: (len(ev), sum(s.elapsed_time(e) for s, e in ev)) for tag, ev in self.events.items()}


timer = None


class region:
    """`with region("dw_fwd"):` brackets the launches inside with HIP events when a timer is installed."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if timer is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if timer is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            timer.record(self.tag, self.start, end)
        return False
