"""Autograd functions that run the Voice100 CNN blocks on the HIP kernels.

Every function here launches kernels of libvoice100_hip.so on torch's current
stream; there is no CPU or stock-PyTorch fallback.  Activations are fp32
[B, C, T]; `precision` selects the matrix-core path of the 1x1 convolutions:
"fp32" (exact fp32 MFMA, the parity path) or "bf16" (operands rounded to bf16
while staging, fp32 accumulate).

Data flow of one InvertedResidual block (reference: voice100/models/asr.py:40-59)
in training mode -- raw conv outputs a1/a2/a3 are the only large tensors written;
BatchNorm + ReLU6 are applied by the *consumer* while it loads its input:

    x --pw GEMM--> a1 (+stats) --dw (BN1+ReLU6 on load)--> a2 (+stats)
      --pw GEMM (BN2+ReLU6 on load)--> a3 (+stats) --affine(+x)--> y
"""
import ctypes
import functools
import weakref
import os
from typing import Optional

import torch

from . import _native as N


@functools.lru_cache(maxsize=None)
def pitch16(T: int, B: int) -> int:
    """Row pitch (elements) of a 16-bit-stored activation tensor [B][C][P]: the library's ONE rule (csrc/common.h v100_pitch16 --
    a multiple of 8; for B > 1 and T >= 256 whole 128-byte lines).  Every allocation of such a tensor goes through here."""
    return int(N.helper("v100_row_pitch16", int(T), int(B)))

# The kernels clamp out-of-range ids instead of faulting (embedding rows: bn.hip; CTC labels -> blank: ctc.hip), where
# the reference's nn.Embedding / nn.CTCLoss raise.  VOICE100_CHECK_IDS=1 validates them on the host first (one device
# sync per call, so it is a debugging switch, off by default).
CHECK_IDS = os.environ.get("VOICE100_CHECK_IDS", "0") not in ("", "0")


def _check_ids(idx: torch.Tensor, n: int, what: str) -> None:
    if CHECK_IDS and idx.numel():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n:
            raise IndexError(f"{what}: index out of range [0, {n}) (min {lo}, max {hi})")

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

_PRECISION = "fp32"
# Storage of the big hidden tensors of a training-mode block under precision "bf16" (include/voice100_hip.h, "act16"):
# 0 = fp32 everywhere, 1 = a1 / a2 (the tensors saved for backward) as bf16, 2 = also the hidden gradients dz2 / dz1.
# The reference under bf16 autocast keeps exactly these tensors in bf16; statistics and accumulators stay fp32 here.
# 3 = also the project output a3 (saved for backward) and its gradient da3: every tensor that is internal to a block.
# 4 = plus a bf16 shadow of each block output for the next block's GEMMs; 5 (default) = the forward residual stream of a stack in that 16-bit form
# only (set_activation_storage)
_ACT16 = int(os.environ.get("VOICE100_ACT16", "5"))


def set_activation_storage(level: int) -> None:
    """0: fp32 activations; 1: saved hidden activations bf16; 2: hidden gradients bf16 as well; 3: also the project output and
    its gradient; 4: plus a bf16 shadow of every block output beside the fp32 tensor, which the next block's expand GEMM and
    expand weight gradient load as their X operand (bf16 precision only; block inputs / outputs themselves stay fp32 up to here);
    5: the forward residual stream of a stack in ONE 16-bit form -- a block's residual is read from that bf16 copy and interior blocks
    of a stack do not write the fp32 copy of their output at all (what the reference's bf16 autocast run keeps) -- and, since round 6,
    the GRADIENT between two residual blocks of one stack call travels as bf16 too where both sides run the finished-gradient kernels
    (what autograd hands back for a bf16 stream; V100_IR_GRAD16=0 keeps it fp32).  Weight gradients, statistics and the gradients
    entering and leaving a stack stay fp32."""
    global _ACT16
    if level not in (0, 1, 2, 3, 4, 5):
        raise ValueError("activation storage level must be 0 ... 5")
    _ACT16 = level


def get_activation_storage() -> int:
    return _ACT16


def set_matmul_precision(p: str) -> None:
    """Operand format of the MFMA GEMMs: "fp32" (default, exact fp32, the parity path), "bf16" (throughput path,
    training and inference) or "fp16" (IEEE half operands, inference only: BASELINE config 5)."""
    global _PRECISION
    if p not in ("fp32", "bf16", "fp16"):
        raise ValueError("precision must be 'fp32', 'bf16' or 'fp16'")
    _PRECISION = p


def _fmt(precision: Optional[str]) -> int:
    """0 fp32, 1 bf16, 2 fp16 -- the `use_bf16` value of the C ABI (truthy for both 16-bit formats)."""
    precision = precision or _PRECISION
    try:
        return {"fp32": 0, "bf16": 1, "fp16": 2}[precision]
    except KeyError:
        raise ValueError(f"unknown precision {precision!r}") from None


def _no_fp16_training(fmt: int, what: str):
    if fmt == 2:
        raise RuntimeError(f"{what}: 'fp16' is an inference precision (no gradient kernels); train with 'bf16' or 'fp32'")


def get_matmul_precision() -> str:
    return _PRECISION


def _f32(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _check(x: torch.Tensor, name: str):
    if not x.is_cuda:
        raise RuntimeError(f"{name}: voice100_amd runs on the GPU only (got a {x.device} tensor); "
                           "there is no CPU fallback -- see oracle/ for the CPU checker")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {x.dtype}")


def conv_out_len(t: int, k: int, stride: int) -> int:
    pad = (k - 1) // 2
    return (t + 2 * pad - k) // stride + 1


class _Weights:
    """fp32 [M, K] weight plus the copies a step needs (16-bit and/or transposed).  fmt: 0 fp32, 1 bf16, 2 fp16."""

    def __init__(self, w2d: torch.Tensor, fmt, transposed: bool):
        fmt = int(fmt)
        self.w = w2d
        m, k = w2d.shape
        self.w_bf = self.wt = self.wt_bf = None
        if fmt == 2:
            _no_fp16_training(2 if transposed else 0, "backward GEMM")
            self.w_bf = torch.empty((m, k), dtype=torch.float16, device=w2d.device)
            N.call("v100_weight_prep_f16", w2d, m, k, self.w_bf)
            return
        if fmt:
            self.w_bf = torch.empty((m, k), dtype=torch.bfloat16, device=w2d.device)
        if transposed:
            if fmt:
                self.wt_bf = torch.empty((k, m), dtype=torch.bfloat16, device=w2d.device)
            else:
                self.wt = _f32(k, m, like=w2d)
        if self.w_bf is not None or self.wt is not None or self.wt_bf is not None:
            N.call("v100_weight_prep", w2d, m, k, self.w_bf, self.wt, self.wt_bf)


# Tensors DERIVED from a parameter -- 16-bit copies, re-laid-out matrices -- kept while the parameter object lives and its version
# counter stands (an optimizer step, load_state_dict or any in-place write bumps it): an inference forward then launches no
# weight-preparation kernel at all, a training step rebuilds exactly what it rebuilt before.  Keyed by the parameter OBJECT (weak
# reference: a new tensor at a recycled address can never hit an old entry), one entry per (parameter, tag).
_DERIVED = {}


def _derived(w: torch.Tensor, tag, build):
    key = (id(w), tag)
    ent = _DERIVED.get(key)
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == w.data_ptr():
        return ent[3]
    val = build()
    try:
        ref = weakref.ref(w, lambda _r, k=key: _DERIVED.pop(k, None))
    except TypeError:
        return val
    _DERIVED[key] = (ref, w._version, w.data_ptr(), val)
    return val


def clear_derived_cache() -> None:
    """Drop every cached derived tensor (16-bit weight copies, re-laid-out matrices).  Needed only after writing parameters in a way that
    bypasses autograd's version counter -- `w.data.copy_(...)`, a foreign kernel on `w.data_ptr()` -- ; everything that goes through torch
    (optimizers, load_state_dict, nn.init under no_grad) or through this package's own kernels is seen without it.  (The eval-mode block
    caches are dropped by module.train() / module.eval().)"""
    _DERIVED.clear()


def _weights_of(w: torch.Tensor, rows: int, cols: int, fmt, transposed: bool) -> "_Weights":
    """_Weights of the parameter w viewed as [rows, cols], through the derived-tensor cache."""
    return _derived(w, ("w", rows, cols, int(fmt), bool(transposed)), lambda: _Weights(w.detach().reshape(rows, cols), fmt, transposed))


def _touched(tensors) -> None:
    """The library has written these module tensors (parameters, BatchNorm running statistics) through raw pointers: advance their autograd
    version counters, which is what every cache keyed on `_version` (the folded eval coefficients, _derived) and autograd's own
    saved-tensor checks rely on.  Host-side metadata only, no launch."""
    torch.autograd.graph.increment_version(tensors)


def _pw_gemm(a_f32, a_bf, x, y, m, k, t, b, bf16, x2=None, xa=None, xb=None, xc=None, x_mode=0, bias=None,
             ea=None, eb=None, r=None, epi=0, stats=None):
    N.call("v100_pw_gemm", a_f32, a_bf, x, x2, xa, xb, xc, x_mode, y, bias, ea, eb, r, epi, stats, b, m, k, t, int(bf16))


def _bn_train(stats, parts, count, bn_w, bn_b, rm, rv, nbt, c, like):
    scale, shift, mean, rstd = (_f32(c, like=like) for _ in range(4))
    N.call("v100_bn_finalize_train", stats, parts, count, bn_w, bn_b, rm, rv, nbt, BN_MOMENTUM, BN_EPS, scale, shift, mean, rstd, c)
    _touched((rm, rv, nbt))
    return scale, shift, mean, rstd


def _bn_eval(bn_w, bn_b, rm, rv, c, like):
    scale, shift = _f32(c, like=like), _f32(c, like=like)
    N.call("v100_bn_eval_coeffs", bn_w, bn_b, rm, rv, BN_EPS, scale, shift, c)
    return scale, shift


def _bn_bwd(partial, parts, count, gamma, mean, rstd, c, like):
    p, q, r, dg, db = (_f32(c, like=like) for _ in range(5))
    N.call("v100_bn_bwd_finalize", partial, parts, count, gamma, mean, rstd, p, q, r, dg, db, c)
    return p, q, r, dg, db


def _ptr_table(items):
    arr = (ctypes.c_void_p * len(items))()
    for i, t in enumerate(items):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class InvertedResidualTrainFn(torch.autograd.Function):
    """Training-mode InvertedResidual (asr.py:40-59): batch statistics, running-stat update, autograd.
    Forward and backward are ONE call each into the library's block executor (csrc/block.hip), which
    sequences the GEMM / depthwise / BatchNorm kernels on the current stream."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, wd, g2, b2, w3, g3, b3, rm1, rv1, nbt1, rm2, rv2, nbt2, rm3, rv3, nbt3,
                kernel_size, stride, use_residual, precision, prep=None, x16=None, want_shadow=False, frozen=False):
        # frozen: the block is in eval() but inside autograd (partial-freeze fine-tuning; an eval-mode model called without no_grad):
        # BatchNorm normalises with its RUNNING statistics, updates nothing, and back-propagates through that fixed affine -- what
        # nn.BatchNorm1d does in eval mode (asr.py:36,52).  Runs on the executor's fp32-storage path (shape[9] bit 3).
        _check(x, "InvertedResidual")
        x = x.contiguous()
        B, cin, T = x.shape
        hid, cout = w1.shape[0], w3.shape[0]
        k = int(kernel_size)
        T2 = conv_out_len(T, k, stride)
        bf16 = _fmt(precision)
        _no_fp16_training(bf16, "InvertedResidual (training mode)")
        # shape[9]: bit 0 = weights prepared by the stack, bit 1 = the pointer table carries the bf16-shadow slots (level 4)
        shadows = bf16 == 1 and _ACT16 >= 4 and not frozen
        shape = (ctypes.c_int * 11)(B, cin, hid, cout, T, k, int(stride), int(bool(use_residual)), int(bf16),
                                    int(prep is not None) | (2 if shadows else 0) | (8 if frozen else 0), 0)
        if bf16 == 1 and _ACT16 and not frozen and N.helper("v100_ir_act16_supported", shape):
            shape[10] = _ACT16
            pitch = pitch16(T, B)                      # bf16 rows are pitched (aligned 8 / 16-byte accesses; long rows on 128-byte lines)
            a1 = torch.empty((B, hid, pitch), dtype=torch.bfloat16, device=x.device)
            a2 = torch.empty((B, hid, pitch), dtype=torch.bfloat16, device=x.device)
            a3 = torch.empty((B, cout, pitch), dtype=torch.bfloat16, device=x.device) if _ACT16 >= 3 else _f32(B, cout, T2, like=x)
        else:
            a1 = _f32(B, hid, T, like=x)
            a2 = _f32(B, hid, T2, like=x)
            a3 = _f32(B, cout, T2, like=x)
        y = _f32(B, cout, T2, like=x)
        # level 4: a bf16 copy of y (pitched rows) for the next block's expand GEMM / expand weight gradient; x16 = the copy
        # of x the previous block wrote (only the act16 executor reads it)
        y16 = torch.empty((B, cout, pitch16(T2, B)), dtype=torch.bfloat16, device=x.device) if (shadows and want_shadow) else None
        if not (shadows and shape[10] >= 4):
            x16 = None
        coef = _f32(12, max(hid, cout), like=x)
        ws = torch.empty(N.helper("v100_ir_fwd_workspace_bytes", shape), dtype=torch.uint8, device=x.device)
        if prep is None:
            prep = torch.empty(N.helper("v100_ir_prep_bytes", shape), dtype=torch.uint8, device=x.device)
        tensors = (x, w1, g1, b1, rm1, rv1, nbt1, wd, g2, b2, rm2, rv2, nbt2, w3, g3, b3, rm3, rv3, nbt3, a1, a2, a3, y, coef, ws, prep)
        if shadows:
            tensors = tensors + (x16, y16)
        for t in tensors[:19]:
            if not t.is_contiguous() or not t.is_cuda:
                raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors")
        N.call("v100_ir_fwd_train", shape, _ptr_table(tensors))
        if not frozen:
            _touched((rm1, rv1, nbt1, rm2, rv2, nbt2, rm3, rv3, nbt3))
        ctx.save_for_backward(x, a1, a2, a3, w1, wd, w3, g1, g2, g3, coef, prep, x16 if x16 is not None else coef)
        ctx.has_x16 = x16 is not None
        ctx.shape = shape
        if y16 is not None:
            ctx.mark_non_differentiable(y16)
        # autograd would otherwise hand backward() a zero-FILLED gradient for the (non-differentiable) shadow output: a 17 MB
        # fill launch per block and step
        ctx.set_materialize_grads(False)
        ctx.y_shape = y.shape
        return y, y16

    @staticmethod
    def backward(ctx, dy, _dy16=None):
        x, a1, a2, a3, w1, wd, w3, g1, g2, g3, coef, prep, x16 = ctx.saved_tensors
        if not ctx.has_x16:
            x16 = None
        shape = ctx.shape
        if dy is None:                     # the block output was not used downstream (materialize_grads is off)
            dy = torch.zeros(ctx.y_shape, dtype=torch.float32, device=x.device)
        dy = dy.contiguous()
        hid, cin = w1.shape[0], w1.shape[1]
        cout, k = w3.shape[0], wd.shape[2]
        # parameter gradients in one allocation: dW1 dg1 db1 dWd dg2 db2 dW3 dg3 db3
        sizes = (hid * cin, hid, hid, hid * k, hid, hid, cout * hid, cout, cout)
        flat = _f32(sum(sizes), like=x)
        parts, off = [], 0
        for n in sizes:
            parts.append(flat[off:off + n])
            off += n
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = torch.empty(N.helper("v100_ir_bwd_workspace_bytes", shape), dtype=torch.uint8, device=x.device)
        extra = (x16,) if (shape[9] & 2) else ()
        N.call("v100_ir_bwd", shape, _ptr_table((x, a1, a2, a3, w1, wd, w3, g1, g2, g3, coef, dy, dx) + tuple(parts) + (ws, prep) + extra))
        dW1, dg1, db1, dWd, dg2, db2, dW3, dg3, db3 = parts
        return (dx, dW1.view_as(w1), dg1, db1, dWd.view_as(wd), dg2, db2, dW3.view_as(w3), dg3, db3) + (None,) * 17


def _block_tensors(blk):
    """The 18 parameter / buffer tensors of an InvertedResidual in the stack executor's order (include/voice100_hip.h).
    Read straight from the modules' _parameters / _buffers dictionaries: this runs once per block and step (the identity check of
    ir_stack_train), and attribute access on an nn.Module goes through Module.__getattr__ (0.4 ms of a step's 2.6 ms of host time)."""
    mods = blk.__dict__["_modules"]["conv"].__dict__["_modules"]
    pw, dw, pl, bn3 = mods["0"].__dict__["_modules"], mods["1"].__dict__["_modules"], mods["2"], mods["3"]
    c1, bn1, cd, bn2 = pw["0"], pw["1"], dw["0"], dw["1"]
    p1, b1, p2, b2, p3, b3 = bn1._parameters, bn1._buffers, bn2._parameters, bn2._buffers, bn3._parameters, bn3._buffers
    return (c1._parameters["weight"], p1["weight"], p1["bias"], b1["running_mean"], b1["running_var"], b1["num_batches_tracked"],
            cd._parameters["weight"], p2["weight"], p2["bias"], b2["running_mean"], b2["running_var"], b2["num_batches_tracked"],
            pl._parameters["weight"], p3["weight"], p3["bias"], b3["running_mean"], b3["running_var"], b3["num_batches_tracked"])


_STACK_PLANS = {}
# data-parallel gradient buffers that want stack gradients written in place (dist.FlatGradBuckets).  WEAK references, newest first:
# a FlatGradBuckets that is simply dropped (a TrainStep recreated for the same model) neither stays alive through this list with its
# flat buffer nor keeps receiving the stack gradients in place of its successor.
_GRAD_ARENAS = []


def register_grad_arena(a) -> None:
    import weakref
    unregister_grad_arena(a)
    _GRAD_ARENAS.insert(0, weakref.ref(a))


def unregister_grad_arena(a) -> None:
    _GRAD_ARENAS[:] = [r for r in _GRAD_ARENAS if r() is not None and r() is not a]


def _grad_arena_for(weights, numel):
    """A registered buffer slice in which `weights` (the stack's trainable tensors, in its gradient order) lie back to back."""
    if any(w.grad is not None for w in weights):
        return None             # gradients are being ACCUMULATED (no begin_step since the last backward): the arena holds the running sum
    for r in _GRAD_ARENAS:
        a = r()
        if a is None or not a._hooks:
            continue            # collected, or its hooks were removed: it no longer takes part in the step
        if weights[0] not in a._view:
            continue            # another model's buffer
        # the newest live buffer that holds these parameters decides alone: when it declines (slice claimed already this step,
        # bucket on the wire, layout mismatch) the gradients take the ordinary path, never an older buffer
        buf = a.grad_arena(weights)
        if buf is not None and buf.numel() != numel:
            a._claimed.discard(a._view[weights[0]][0])
            buf = None
        return buf
    return None
_STACK_SEGMENT = None          # blocks per autograd node of a stack; None = automatic (ir_stack_train)


def set_stack_segment(n: Optional[int]) -> None:
    """Blocks per autograd node of the stack executor: None = automatic (the whole run in a single-process job, three under data
    parallelism); an int forces it (bench.py measures the data-parallel form, 3, on one GPU)."""
    global _STACK_SEGMENT
    if n is not None and n < 1:
        raise ValueError("segment must be >= 1 or None")
    _STACK_SEGMENT = n


def _stack_plan(cfgs, B, T, bf16, level, last_shadow):
    """(desc ctypes array, per-block offsets, totals) of a run of blocks at this input shape: computed once per distinct key."""
    key = (cfgs, B, T, bf16, level, last_shadow)
    ent = _STACK_PLANS.get(key)
    if ent is None:
        n = len(cfgs)
        desc = (ctypes.c_int * (6 + 6 * n))(n, B, T, bf16, level, int(last_shadow), *[v for c in cfgs for v in c])
        plan = (ctypes.c_longlong * (8 * n + 6))()
        if N.helper("v100_ir_stack_plan", desc, plan) < 0:
            raise RuntimeError("InvertedResidual stack: invalid shape")
        if len(_STACK_PLANS) > 256:                 # time-stretched lengths: at most ~100 distinct T per run
            _STACK_PLANS.clear()
        ent = _STACK_PLANS[key] = (desc, [tuple(plan[8 * i:8 * i + 8]) for i in range(n)], tuple(plan[8 * n:8 * n + 6]))
    return ent


class IRStackTrainFn(torch.autograd.Function):
    """A run of consecutive training-mode InvertedResidual blocks (asr.py:67-76, tts.py:17-25, 72-76) as ONE autograd node:
    one host call into the library's stack executor per direction (csrc/block.hip), every activation kept for backward in one
    allocation.  Same kernels in the same order as the per-block InvertedResidualTrainFn: bit-identical results."""

    @staticmethod
    def forward(ctx, x, x16, meta, *weights):
        # weights: the 9 trainable tensors of each block (w1 g1 b1 wd g2 b2 w3 g3 b3) -- the autograd inputs; meta carries the full
        # 18-tensor table per block (running statistics included: buffers, no gradient, so not autograd inputs -- half the arguments)
        cfgs, precision, last_shadow, params = meta
        _check(x, "InvertedResidual stack")
        x = x.contiguous()
        B, cin, T = x.shape
        n = len(cfgs)
        if len(params) != 18 * n or len(weights) != 9 * n or cfgs[0][0] != cin:
            raise RuntimeError("InvertedResidual stack: parameter list / input width do not match the block configuration")
        bf16 = _fmt(precision)
        _no_fp16_training(bf16, "InvertedResidual (training mode)")
        level = _ACT16 if bf16 == 1 else 0
        desc, blocks, totals = _stack_plan(cfgs, B, T, bf16, level, bool(last_shadow))
        if not (bf16 == 1 and level >= 4):
            x16 = None
        blob = torch.empty(totals[0], dtype=torch.uint8, device=x.device)
        ptab = (ctypes.c_void_p * (18 * n))(*[t.data_ptr() for t in params])
        N.call("v100_ir_stack_fwd_train", desc, ptab, x, x16, blob)
        _touched([params[18 * i + j] for i in range(n) for j in (3, 4, 5, 9, 10, 11, 15, 16, 17)])     # the blocks' running statistics
        o = blocks[-1]
        cout, T2 = cfgs[-1][2], o[7]
        y = blob[o[3]:o[3] + 4 * B * cout * T2].view(torch.float32).view(B, cout, T2)
        y16 = None
        if o[4] >= 0:
            P2 = pitch16(T2, B)
            y16 = blob[o[4]:o[4] + 2 * B * cout * P2].view(torch.bfloat16).view(B, cout, P2)
            ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)
        # (plain attributes, not save_for_backward: y is a view of the blob, and the fused optimiser updates parameters through raw
        # pointers anyway -- tensor version counters say nothing here)
        ctx.blob, ctx.x, ctx.x16, ctx.params, ctx.desc, ctx.totals, ctx.cfgs = blob, x, x16, params, desc, totals, cfgs
        ctx.ptab = ptab                                   # the same 18 n pointers serve this forward's backward (same step, same storages)
        ctx.y_shape = y.shape
        return y, y16

    @staticmethod
    def backward(ctx, dy, _dy16=None):
        x, params, cfgs, totals = ctx.x, ctx.params, ctx.cfgs, ctx.totals
        if ctx.blob is None:
            raise RuntimeError("InvertedResidual stack: backward ran already for this forward (its activations are freed as soon as "
                               "they are consumed; retain_graph / a second backward needs a second forward)")
        n = len(cfgs)
        if dy is None:
            dy = torch.zeros(ctx.y_shape, dtype=torch.float32, device=x.device)
        dy = dy.contiguous()
        # under data parallelism the gradients are written straight into the exchange buffer (no packing copy): same values, another home
        weights = [params[18 * b + j] for b in range(n) for j in (0, 1, 2, 6, 7, 8, 12, 13, 14)]
        grads = _grad_arena_for(weights, totals[2]) if _GRAD_ARENAS else None
        if grads is None:
            grads = _f32(totals[2], like=x)
        ws = torch.empty(totals[1], dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        N.call("v100_ir_stack_bwd", ctx.desc, ctx.ptab, x, ctx.x16, ctx.blob, dy, dx, grads, ws)
        ctx.blob = None                                   # the activations are dead: free them before the rest of backward runs
        out, off = [], 0
        for i, (cin, hid, cout, k, _, _) in enumerate(cfgs):
            p = params[18 * i:18 * i + 18]
            for j, numel in ((0, hid * cin), (1, hid), (2, hid), (6, hid * k), (7, hid), (8, hid), (12, cout * hid), (13, cout), (14, cout)):
                out.append(grads[off:off + numel].view_as(p[j]))
                off += numel
        return (dx, None, None) + tuple(out)


def _stack_segments(blocks, segment):
    """[(cfgs, params tuple, blocks)] per segment: the static part of ir_stack_train's work, cached on the run's first block (it dies
    with the module; module attributes and parameter objects are static -- a parameter that is REPLACED rather than updated in place is
    caught by the identity check in ir_stack_train)."""
    ent = blocks[0].__dict__.get("_v100_stack_meta")
    if ent is None or len(ent[0]) != len(blocks) or any(a is not b for a, b in zip(ent[0], blocks)):
        ent = blocks[0].__dict__["_v100_stack_meta"] = (tuple(blocks), {})
    segs = ent[1].get(segment)
    if segs is None:
        segs = []
        for s in range(0, len(blocks), segment):
            seg = blocks[s:s + segment]
            cfgs = tuple((b.conv[0][0].in_channels, b.conv[0][0].out_channels, b.conv[2].out_channels, int(b.kernel_size), int(b.stride),
                          int(bool(b.use_residual))) for b in seg)
            params = tuple(t for b in seg for t in _block_tensors(b))
            for t in params:
                if not t.is_cuda or not t.is_contiguous():
                    raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors (no CPU fallback)")
            segs.append((cfgs, params, seg))
        ent[1][segment] = segs
    return segs


def _stack_eligible(blocks) -> bool:
    """The stack executor runs every block of a run in training mode and bypasses Module.__call__: it may only stand in for
    `nn.Sequential` dispatch (asr.py:76, tts.py:25) when that is what per-module dispatch would have done -- every block in training
    mode (a block the user froze with block.eval() keeps its running statistics and uses them) and no forward / pre-forward hook
    registered on any block (the hooks would be skipped).  Global module hooks count as hooks on every block."""
    import torch.nn.modules.module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return False
    for b in blocks:
        if not b.training or b._forward_hooks or b._forward_pre_hooks:
            return False
        # a frozen sub-module (block.conv[1][1].eval(): one BatchNorm on running statistics) is not expressible by the fused block
        # either way; the per-block path sees the same block.training flag, so it is no reason to leave the stack
    return True


def ir_stack_train(blocks, x, precision: Optional[str] = None, segment: Optional[int] = None):
    """Training-mode forward of consecutive InvertedResidual modules through the stack executor.  `segment` = blocks per autograd
    node: None -> the whole run as one node in a single-process job, three blocks per node under data parallelism (the gradient
    buckets of the later blocks are then all-reduced while the earlier blocks' backward still runs, voice100_amd/dist.py).
    Falls back to calling the modules one by one (what the reference's nn.Sequential does) when any block is in eval mode
    (partial-freeze fine-tuning: that block then runs with frozen BatchNorm statistics and still back-propagates) or carries forward hooks."""
    blocks = list(blocks)
    if not _stack_eligible(blocks):
        for b in blocks:          # (an eval-mode block that takes part in autograd runs with FROZEN statistics: layers.InvertedResidual.forward)
            x = b(x)
        return x
    if segment is None:
        segment = _STACK_SEGMENT
    if segment is None:
        import torch.distributed as dist
        segment = 3 if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else len(blocks)
    precision = precision or _PRECISION
    segs = _stack_segments(blocks, segment)
    for i, (cfgs, params, seg) in enumerate(segs):
        # any parameter / buffer object that was swapped out since the tuple was cached (Module._apply replaces BatchNorm buffers on
        # .to() / .float(); load_state_dict(assign=True) and module._parameters[...] = new replace parameters) invalidates it: the
        # executor would otherwise read dead weights and write running statistics into dead buffers.  18 identity checks per block.
        current = tuple(t for b in seg for t in _block_tensors(b))
        if any(a is not c for a, c in zip(params, current)):
            for t in current:
                if not t.is_cuda or not t.is_contiguous():
                    raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors (no CPU fallback)")
            params = current
            segs[i] = (cfgs, params, seg)
        sh = getattr(x, "_v100_shadow", None)
        x16 = sh[0] if (sh is not None and sh[1] == x._version and sh[0].shape[:2] == x.shape[:2]
                        and sh[0].shape[2] == pitch16(x.shape[2], x.shape[0])) else None
        last = i + 1 == len(segs)
        weights = [params[18 * b + j] for b in range(len(cfgs)) for j in (0, 1, 2, 6, 7, 8, 12, 13, 14)]
        y, y16 = IRStackTrainFn.apply(x, x16, (cfgs, precision, not last, params), *weights)
        if y16 is not None:
            y._v100_shadow = (y16, y._version)
        x = y
    return x


def prepare_block_weights(blocks, precision: Optional[str] = None) -> None:
    """bf16 / transposed copies of the two 1x1 weights of every InvertedResidual in `blocks`, in ONE launch
    (v100_ir_prep_batched), into slices of ONE freshly allocated buffer that the blocks' next forward hands to the
    executor and saves for its backward.  Called by the stacks (ConvVoiceEncoder, VoiceDecoder, ...) at the top of EVERY
    training-mode forward: nothing is cached or overwritten across forwards (weights change under the optimiser, and a
    backward that runs after a later forward must still see the copies of ITS forward).  A block that runs without this call prepares its own copies as before."""
    bf16 = _fmt(precision)
    _no_fp16_training(bf16, "InvertedResidual (training mode)")
    todo, total = [], 0
    for blk in blocks:
        w1, w3 = blk.conv[0][0].weight, blk.conv[2].weight
        if not w1.is_cuda:
            raise RuntimeError("InvertedResidual: parameters must be CUDA tensors (no CPU fallback)")
        hid, cin = w1.shape[0], w1.shape[1]
        cout = w3.shape[0]
        shape = (0, cin, hid, cout, 0, 0, 1, 0, int(bf16), 1, 0)
        nbytes = N.helper("v100_ir_prep_bytes", (ctypes.c_int * 11)(*shape))
        todo.append((blk, shape, w1, w3, total, nbytes))
        total += (nbytes + 255) // 256 * 256
    if not todo:
        return
    pool = torch.empty(total, dtype=torch.uint8, device=todo[0][2].device)
    bufs = [pool[off:off + nbytes] for _, _, _, _, off, nbytes in todo]
    for i in range(0, len(todo), 32):
        chunk = todo[i:i + 32]
        shapes = (ctypes.c_int * (11 * len(chunk)))(*[v for c in chunk for v in c[1]])
        N.call("v100_ir_prep_batched", shapes, _ptr_table([c[2] for c in chunk]), _ptr_table([c[3] for c in chunk]),
               _ptr_table(bufs[i:i + 32]), len(chunk))
    for (blk, _, _, _, _, _), buf in zip(todo, bufs):
        blk._prep_buf = buf
        blk._prep_fresh = bf16          # consumed (once) by the block's next training-mode forward at this precision


def prepared_weights_of(blk, precision: Optional[str] = None):
    """The block's prepared-weights buffer if prepare_block_weights filled it for THIS forward, else None."""
    bf16 = _fmt(precision)
    fresh = getattr(blk, "_prep_fresh", None)
    if fresh is not None and fresh == bf16:
        buf, blk._prep_fresh, blk._prep_buf = blk._prep_buf, None, None     # the forward's ctx now owns it
        return buf
    return None


def inverted_residual_eval_cached(blk, x, precision: Optional[str] = None):
    """Eval-mode InvertedResidual through the block executor: the folded BatchNorm coefficients and bf16 weight copies
    are cached on the module (refilled when any parameter / running statistic changes: tensor versions are the key),
    so a forward is one host call = 3 kernel launches.  Inference only (no autograd)."""
    _check(x, "InvertedResidual")
    x = x.contiguous()
    pw, dw, pl, bn3 = blk.conv[0], blk.conv[1], blk.conv[2], blk.conv[3]
    bn1, bn2 = pw[1], dw[1]
    w1, wd, w3 = pw[0].weight, dw[0].weight, pl.weight
    bf16 = _fmt(precision)
    B, cin, T = x.shape
    hid, cout, k = w1.shape[0], w3.shape[0], int(blk.kernel_size)
    T2 = conv_out_len(T, k, blk.stride)
    shape = (ctypes.c_int * 11)(B, cin, hid, cout, T, k, int(blk.stride), int(bool(blk.use_residual)), int(bf16), 0, 0)
    params = (w1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
              w3, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var)
    key = (bf16,) + tuple((t.data_ptr(), t._version) for t in params) + (wd.data_ptr(), wd._version)
    if getattr(blk, "_eval_key", None) != key:
        for t in params + (wd,):
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors (no CPU fallback)")
        cache = torch.empty(N.helper("v100_ir_eval_cache_bytes", shape), dtype=torch.uint8, device=x.device)
        N.call("v100_ir_eval_prep", shape, _ptr_table(tuple(t.detach() for t in params) + (cache,)))
        blk._eval_cache, blk._eval_key = cache, key
    y = _f32(B, cout, T2, like=x)
    if bf16 == 1 and _ACT16 and N.helper("v100_ir_act16_supported", shape):
        # precision "bf16": the two hidden tensors (4x the block's width, values in [0, 6] after BatchNorm + ReLU6) stored as bf16 with
        # pitched rows -- half the bytes of the big streams of all three kernels (the GEMMs round them to bf16 as operands anyway)
        shape[10] = 1
        pitch = pitch16(T, B)
        h1 = torch.empty((B, hid, pitch), dtype=torch.bfloat16, device=x.device)
        h2 = torch.empty((B, hid, pitch), dtype=torch.bfloat16, device=x.device)
    else:
        h1, h2 = _f32(B, hid, T, like=x), _f32(B, hid, T2, like=x)
    N.call("v100_ir_fwd_eval", shape, _ptr_table((x, w1.detach(), wd.detach(), w3.detach(), blk._eval_cache, h1, h2, y)))
    return y


# ---- channel-major inference (round 4) ------------------------------------------------------------------------------
# Activations [C][B][P] (P = pitch16(T, B)): one [C x (B P)] matrix per tensor, the utterances' rows back to back.  Each 1x1
# convolution of a block is then ONE GEMM over all B P columns, the depthwise kernel walks a channel's rows contiguously and packs
# short rows several to a wave item (include/voice100_hip.h).  Used by the models' eval-mode forwards at the 16-bit precisions.
EVAL_CM = os.environ.get("VOICE100_EVAL_CM", "1") not in ("", "0")


def eval_cm_supported(blocks, T: int, precision: Optional[str] = None, batch: int = 1) -> bool:
    """True when a run of eval-mode InvertedResidual blocks can run channel-major: 16-bit precision, stride 1, kernel sizes with a
    matrix-pipe depthwise kernel, rows that fit one wave item, nothing that Module.__call__ would have to do (hooks), and a batch
    inside the kernels' index ranges (v100_cm_to_btc puts the batch in grid.z: B <= 65535; v100_ir_fwd_eval addresses the
    [C][B * P] matrix with 32-bit byte offsets: B * P <= 0x7fffff00 -- past either the per-module path runs instead of an error)."""
    import torch.nn.modules.module as _m
    if not EVAL_CM or _fmt(precision) == 0 or T > 768 or T < 1:
        return False
    if batch > 65535 or batch * pitch16(T, batch) > 0x7fffff00:
        return False
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return False
    for b in blocks:
        if b.training or b.stride != 1 or b._forward_hooks or b._forward_pre_hooks:
            return False
        if not N.helper("v100_dw_mfma_supported", int(b.kernel_size), 1):
            return False
        w1 = b.conv[0][0].weight
        if not w1.is_cuda or (w1.shape[0] % 2) or (w1.shape[1] % 2) or (b.conv[2].weight.shape[0] % 2):
            return False
    return True


def bct_to_cm(x: torch.Tensor) -> torch.Tensor:
    """[B, C, T] fp32 -> channel-major [C, B * P] fp32 (padding columns zeroed)."""
    _check(x, "bct_to_cm")
    x = x.contiguous()
    B, C, T = x.shape
    P = pitch16(T, B)
    y = _f32(C, B * P, like=x)
    N.call("v100_bct_to_cm", x, y, B, C, T)
    return y


def cm_to_btc(x: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """channel-major [C, B * P] fp32 -> [B, T, C] (the model-edge transpose, asr.py:114)."""
    C = x.shape[0]
    y = _f32(B, T, C, like=x)
    N.call("v100_cm_to_btc", x, y, B, C, T)
    return y


def inverted_residual_eval_cm(blk, x: torch.Tensor, B: int, T: int, precision: Optional[str] = None) -> torch.Tensor:
    """Eval-mode InvertedResidual (stride 1) on a channel-major activation x [cin, B * P]: three launches (one GEMM over all
    columns, the depthwise stage, one GEMM), hidden tensors stored in the GEMMs' 16-bit operand format.  Inference only."""
    pw, dw, pl, bn3 = blk.conv[0], blk.conv[1], blk.conv[2], blk.conv[3]
    bn1, bn2 = pw[1], dw[1]
    w1, wd, w3 = pw[0].weight, dw[0].weight, pl.weight
    fmt = _fmt(precision)
    hid, cin, cout, k = w1.shape[0], w1.shape[1], w3.shape[0], int(blk.kernel_size)
    P = pitch16(T, B)
    if x.shape != (cin, B * P):
        raise RuntimeError("inverted_residual_eval_cm: x must be [cin, B * pitch(T)]")
    shape = (ctypes.c_int * 11)(B, cin, hid, cout, T, k, 1, int(bool(blk.use_residual)), int(fmt), 0, 0)
    params = (w1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
              w3, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var)
    key = (fmt,) + tuple((t.data_ptr(), t._version) for t in params) + (wd.data_ptr(), wd._version)
    if getattr(blk, "_eval_key", None) != key:
        for t in params + (wd,):
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors (no CPU fallback)")
        cache = torch.empty(N.helper("v100_ir_eval_cache_bytes", shape), dtype=torch.uint8, device=x.device)
        N.call("v100_ir_eval_prep", shape, _ptr_table(tuple(t.detach() for t in params) + (cache,)))
        blk._eval_cache, blk._eval_key = cache, key
    shape[10] = 2
    h = torch.empty((2, hid, B * P), dtype=torch.float16 if fmt == 2 else torch.bfloat16, device=x.device)
    y = _f32(cout, B * P, like=x)
    N.call("v100_ir_fwd_eval", shape, _ptr_table((x, w1.detach(), wd.detach(), w3.detach(), blk._eval_cache, h[0], h[1], y)))
    return y


class _EvalStackPlan:
    __slots__ = ("key", "n", "shapes", "ptrs", "caches", "hid_max", "cout_max", "cout_last", "keep")


# Plans live beside the modules, not inside them: a plan holds ctypes pointer arrays, which can be neither pickled nor
# deep-copied, and a module must stay copy.deepcopy()-able / torch.save()-able after an eval forward.
_EVAL_STACK_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def ir_stack_eval_cm(blocks, x: torch.Tensor, B: int, T: int, precision: Optional[str] = None) -> torch.Tensor:
    """A run of eval-mode stride-1 InvertedResidual blocks on a channel-major activation x [cin, B * P]: inverted_residual_eval_cm
    block after block, as ONE call into the library (v100_ir_stack_fwd_eval).  At the configs' own sizes (1-second chunks, B = 2 x 256
    frames) the forward is bound by the host's cost per block -- parameter lookups through nn.Module, two allocations, a pointer table,
    a ctypes call: ~28 us a block against ~20 us of GPU work -- so the shapes, folded-BatchNorm caches and the constant pointers live
    in a plan that is rebuilt only when a parameter or buffer object, its version, the batch, the length or the precision changes."""
    fmt = _fmt(precision)
    n = len(blocks)
    P = pitch16(T, B)
    tens = [_block_tensors(b) for b in blocks]
    # data_ptr() is part of the key: `p.data = other` keeps id() and _version but moves the storage the plan's pointers name
    key = (fmt, B, T, x.device.index, n) + tuple((id(t), t.data_ptr(), t._version) for bt in tens for t in bt)
    plan = _EVAL_STACK_PLANS.get(blocks[0])
    if plan is None or plan.key != key:
        plan = _EvalStackPlan()
        plan.key, plan.n = key, n
        plan.shapes = (ctypes.c_int * (11 * n))()
        plan.ptrs = (ctypes.c_void_p * (8 * n))()
        plan.caches, plan.keep = [], []
        plan.hid_max = plan.cout_max = 0
        cin_prev = x.shape[0]
        for i, (blk, bt) in enumerate(zip(blocks, tens)):
            (w1, g1, b1, rm1, rv1, _n1, wd, g2, b2, rm2, rv2, _n2, w3, g3, b3, rm3, rv3, _n3) = bt
            hid, cin, cout, k = w1.shape[0], w1.shape[1], w3.shape[0], int(blk.kernel_size)
            if cin != cin_prev or int(blk.stride) != 1:
                raise RuntimeError("ir_stack_eval_cm: blocks must chain (cin == previous cout) at stride 1")
            cin_prev = cout
            params = (w1, g1, b1, rm1, rv1, g2, b2, rm2, rv2, w3, g3, b3, rm3, rv3)
            for t in params + (wd,):
                if not t.is_cuda or not t.is_contiguous():
                    raise RuntimeError("InvertedResidual: parameters and buffers must be contiguous CUDA tensors (no CPU fallback)")
            shape = (ctypes.c_int * 11)(B, cin, hid, cout, T, k, 1, int(bool(blk.use_residual)), int(fmt), 0, 0)
            cache = torch.empty(N.helper("v100_ir_eval_cache_bytes", shape), dtype=torch.uint8, device=x.device)
            N.call("v100_ir_eval_prep", shape, _ptr_table(tuple(t.detach() for t in params) + (cache,)))
            shape[10] = 2
            for j in range(11):
                plan.shapes[11 * i + j] = shape[j]
            plan.ptrs[8 * i + 1], plan.ptrs[8 * i + 2], plan.ptrs[8 * i + 3] = w1.data_ptr(), wd.data_ptr(), w3.data_ptr()
            plan.ptrs[8 * i + 4] = cache.data_ptr()
            plan.caches.append(cache)
            plan.keep.append(bt)        # every keyed tensor stays alive with the plan: id() of a replaced parameter can then never be reused by its successor
            plan.hid_max, plan.cout_max = max(plan.hid_max, hid), max(plan.cout_max, cout)
            plan.cout_last = cout
        _EVAL_STACK_PLANS[blocks[0]] = plan
    if x.shape != (plan.shapes[1], B * P):
        raise RuntimeError("ir_stack_eval_cm: x must be [cin, B * pitch(T)]")
    cols = B * P
    h = torch.empty((2, plan.hid_max, cols), dtype=torch.float16 if fmt == 2 else torch.bfloat16, device=x.device)
    out = _f32(plan.cout_last, cols, like=x)
    tmp = _f32(2, plan.cout_max, cols, like=x) if n > 1 else None
    h0, h1 = h.data_ptr(), h[1].data_ptr()
    cur = x.data_ptr()
    ptrs = plan.ptrs
    for i in range(n):
        y = out.data_ptr() if i == n - 1 else tmp[i & 1].data_ptr()
        ptrs[8 * i], ptrs[8 * i + 5], ptrs[8 * i + 6], ptrs[8 * i + 7] = cur, h0, h1, y
        cur = y
    N.call("v100_ir_stack_fwd_eval", n, plan.shapes, ptrs)
    return out


def pointwise_conv1d_cm(x: torch.Tensor, w: torch.Tensor, bias, precision: Optional[str] = None) -> torch.Tensor:
    """nn.Conv1d(kernel_size=1) on a channel-major activation [cin, N] -> [cout, N] (inference): one GEMM over all columns."""
    cout, cin = w.shape[0], w.shape[1]
    n = x.shape[1]
    fmt = _fmt(precision)
    W = _weights_of(w, cout, cin, fmt, False)
    y = _f32(cout, n, like=x)
    _pw_gemm(W.w, W.w_bf, x, y, cout, cin, n, 1, fmt, bias=bias.detach() if bias is not None else None, epi=0)
    return y


def inverted_residual_eval(x, w1, g1, b1, rm1, rv1, wd, g2, b2, rm2, rv2, w3, g3, b3, rm3, rv3,
                           kernel_size, stride, use_residual, precision):
    """Eval-mode InvertedResidual: BatchNorm folded to per-channel scale/shift inside the three kernels."""
    _check(x, "InvertedResidual")
    x = x.contiguous()
    B, cin, T = x.shape
    hid, cout = w1.shape[0], w3.shape[0]
    k = int(kernel_size)
    pad = (k - 1) // 2
    T2 = conv_out_len(T, k, stride)
    bf16 = _fmt(precision)
    W1 = _Weights(w1.detach().reshape(hid, cin), bf16, False)
    W3 = _Weights(w3.detach().reshape(cout, hid), bf16, False)
    s1, t1 = _bn_eval(g1, b1, rm1, rv1, hid, x)
    s2, t2 = _bn_eval(g2, b2, rm2, rv2, hid, x)
    s3, t3 = _bn_eval(g3, b3, rm3, rv3, cout, x)
    h1 = _f32(B, hid, T, like=x)
    _pw_gemm(W1.w, W1.w_bf, x, h1, hid, cin, T, B, bf16, ea=s1, eb=t1, epi=2)
    h2 = _f32(B, hid, T2, like=x)
    G = N.helper("v100_dw_num_groups", B, hid)
    N.call("v100_dwconv", h1, None, wd.detach().reshape(hid, k).contiguous(), None, None, None, 0, h2, None, s2, t2, 1, None, G,
           B, hid, T, T2, k, stride, pad, 0, 1, 0)
    y = _f32(B, cout, T2, like=x)
    _pw_gemm(W3.w, W3.w_bf, h2, y, cout, hid, T2, B, bf16, ea=s3, eb=t3, r=x if use_residual else None, epi=3)
    return y


class PointwiseConvFn(torch.autograd.Function):
    """nn.Conv1d(kernel_size=1, bias optional) on the GEMM kernel (asr.py:91; tts.py:26,77)."""

    @staticmethod
    def forward(ctx, x, w, bias, precision):
        _check(x, "pointwise_conv1d")
        x = x.contiguous()
        B, cin, T = x.shape
        cout = w.shape[0]
        bf16 = _fmt(precision)
        W = _weights_of(w, cout, cin, bf16, False)
        y = _f32(B, cout, T, like=x)
        _pw_gemm(W.w, W.w_bf, x, y, cout, cin, T, B, bf16, bias=bias.detach() if bias is not None else None, epi=0)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bf16 = bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, cin, T = x.shape
        cout = w.shape[0]
        bf16 = ctx.bf16
        _no_fp16_training(bf16, "backward")
        W = _weights_of(w, cout, cin, bf16, True)
        S = N.helper("v100_pw_wgrad_splits", B, cout, cin)
        partial = _f32(S, cout, cin, like=x)
        dW = _f32(cout, cin, like=x)
        N.call("v100_pw_wgrad", dy, None, None, None, None, 0, x, None, None, 0, partial, dW, S, B, cout, cin, T, int(bf16))
        db = None
        if ctx.has_bias:
            G = N.helper("v100_dw_num_groups", B, cout)
            part = _f32(G, cout, 2, like=x)
            N.call("v100_chan_reduce2", dy, None, part, G, B, cout, T)
            db = _f32(cout, like=x)
            N.call("v100_slab_sum0", part, G, db, cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _f32(B, cin, T, like=x)
            _pw_gemm(W.wt, W.wt_bf, dy, dx, cin, cout, T, B, bf16, epi=0)
        return dx, dW.view_as(w), db, None


def pointwise_conv1d(x, w, bias=None, precision: Optional[str] = None):
    return PointwiseConvFn.apply(x, w, bias, precision or _PRECISION)


class TransposeLast2Fn(torch.autograd.Function):
    """torch.transpose(x, 1, 2) materialised ([B,R,C] -> [B,C,R]); asr.py:111,114."""

    @staticmethod
    def forward(ctx, x):
        _check(x, "transpose_last2")
        x = x.contiguous()
        B, R, C = x.shape
        y = _f32(B, C, R, like=x)
        N.call("v100_transpose_last2", x, y, B, R, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not dy.is_contiguous() and dy.transpose(1, 2).is_contiguous():
            return dy.transpose(1, 2)          # written in this op's INPUT layout by its producer (ctc_loss, grad_bvt): a view, no launch
        dy = dy.contiguous()
        B, C, R = dy.shape
        dx = _f32(B, R, C, like=dy)
        N.call("v100_transpose_last2", dy, dx, B, C, R)
        return dx


def tag_transposed(x: torch.Tensor, xt: torch.Tensor):
    """Attach the [B, C, R] twin a kernel wrote beside x [B, R, C] (the augmentation pass, audio.py); transpose_last2() hands it out
    instead of launching -- as long as x has not been written to since and no gradient is wanted through it."""
    x._v100_T = (xt, x._version)


def transpose_last2(x):
    tag = getattr(x, "_v100_T", None)
    if tag is not None and tag[1] == x._version and not x.requires_grad and tag[0].device == x.device \
            and tag[0].shape == (x.shape[0], x.shape[2], x.shape[1]):
        return tag[0]
    y = TransposeLast2Fn.apply(x)
    if y.requires_grad:
        y._v100_grad_T = True      # a loss kernel that consumes y directly may write d loss / d y in x's layout (ctc_loss does)
    return y


class DropoutMaskFn(torch.autograd.Function):
    """y = x * keep / (1 - p) with a caller-supplied keep mask (nn.Dropout(0.2), asr.py:90)."""

    @staticmethod
    def forward(ctx, x, keep, scale):
        x = x.contiguous()
        y = torch.empty_like(x)
        N.call("v100_mul_scale", x, keep, float(scale), y, x.numel())
        ctx.save_for_backward(keep)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        N.call("v100_mul_scale", dy, keep, ctx.scale, dx, dy.numel())
        return dx, None, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in training mode as one kernel each way (v100_dropout_fwd / _bwd): the keep mask is generated in the
    kernel from a 64-bit seed drawn from torch's CPU generator (so torch.manual_seed makes a run repeatable) and kept as
    one byte per element for backward."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _check(x, "dropout")
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        N.call("v100_dropout_fwd", x, int(seed), float(p), y, mask, x.numel())
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        N.call("v100_dropout_bwd", dy, mask, ctx.p, dx, dy.numel())
        return dx, None, None


class DropoutPointwiseFn(torch.autograd.Function):
    """nn.Dropout(p) then nn.Conv1d(kernel_size=1) (LinearCharDecoder, asr.py:85-94) as ONE autograd node: forward is DropoutFn's
    kernel followed by PointwiseConvFn's GEMM; backward takes the weight / bias gradients from the saved dropped input and applies the
    keep mask in the data-gradient GEMM's own epilogue (v100_pw_gemm_dropmask) instead of a pass over the [B, C, T] gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, p, seed, precision):
        _check(x, "dropout_pointwise_conv1d")
        x = x.contiguous()
        B, cin, T = x.shape
        cout = w.shape[0]
        bf16 = _fmt(precision)
        xd = torch.empty_like(x)
        mask = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        N.call("v100_dropout_fwd", x, int(seed), float(p), xd, mask, x.numel())
        W = _weights_of(w, cout, cin, bf16, False)
        y = _f32(B, cout, T, like=x)
        _pw_gemm(W.w, W.w_bf, xd, y, cout, cin, T, B, bf16, bias=bias.detach() if bias is not None else None, epi=0)
        ctx.save_for_backward(xd, w, mask)
        ctx.has_bias = bias is not None
        ctx.bf16 = bf16
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, w, mask = ctx.saved_tensors
        dy = dy.contiguous()
        B, cin, T = xd.shape
        cout = w.shape[0]
        bf16 = ctx.bf16
        _no_fp16_training(bf16, "backward")
        W = _weights_of(w, cout, cin, bf16, True)
        S = N.helper("v100_pw_wgrad_splits", B, cout, cin)
        partial = _f32(S, cout, cin, like=xd)
        dW = _f32(cout, cin, like=xd)
        N.call("v100_pw_wgrad", dy, None, None, None, None, 0, xd, None, None, 0, partial, dW, S, B, cout, cin, T, int(bf16))
        db = None
        if ctx.has_bias:
            G = N.helper("v100_dw_num_groups", B, cout)
            part = _f32(G, cout, 2, like=xd)
            N.call("v100_chan_reduce2", dy, None, part, G, B, cout, T)
            db = _f32(cout, like=xd)
            N.call("v100_slab_sum0", part, G, db, cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _f32(B, cin, T, like=xd)
            if N.helper("v100_pw_gemm_dropmask_supported", B, cin, cout, T, int(bf16)):
                N.call("v100_pw_gemm_dropmask", W.wt, W.wt_bf, dy, dx, mask, ctx.p, B, cin, cout, T, int(bf16))
            else:
                dz = _f32(B, cin, T, like=xd)
                _pw_gemm(W.wt, W.wt_bf, dy, dz, cin, cout, T, B, bf16, epi=0)
                N.call("v100_dropout_bwd", dz, mask, ctx.p, dx, dx.numel())
        return dx, dW.view_as(w), db, None, None, None


def dropout_pointwise_conv1d(x, w, bias, p: float, precision: Optional[str] = None):
    """Training-mode Dropout(p) -> Conv1d(k=1) with the in-kernel mask generator (0 < p < 1)."""
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())        # CPU generator: no device sync (as dropout())
    return DropoutPointwiseFn.apply(x, w, bias, float(p), seed, precision or _PRECISION)


def dropout(x, p: float, training: bool, keep: Optional[torch.Tensor] = None):
    """Inverted dropout. `keep` (0/1 float mask) can be injected for reproducible parity runs; otherwise the mask comes
    from the in-kernel generator."""
    if not training or p == 0.0:
        return x
    if p >= 1.0:                       # nn.Dropout(p=1): all zeros (and a zero gradient); never on the reference's path (p = 0.2)
        return x * 0.0
    if keep is not None:
        return DropoutMaskFn.apply(x, keep, 1.0 / (1.0 - p))
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())        # CPU generator: no device sync
    return DropoutFn.apply(x, p, seed)


class EmbeddingBCTFn(torch.autograd.Function):
    """nn.Embedding followed by transpose(1,2): idx [B,T] int64 -> [B,C,T] (tts.py:81-83,176-177)."""

    @staticmethod
    def forward(ctx, idx, table):
        if not idx.is_cuda:
            raise RuntimeError("embedding: voice100_amd runs on the GPU only")
        idx = idx.contiguous().to(torch.int64)
        B, T = idx.shape
        V, C = table.shape
        _check_ids(idx, V, "embedding")
        out = _f32(B, C, T, like=table)
        N.call("v100_embedding_bct", idx, table.detach().contiguous(), out, B, V, C, T)
        ctx.save_for_backward(idx)
        ctx.vc = (V, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        V, C = ctx.vc
        dy = dy.contiguous()
        B, _, T = dy.shape
        dtable = _f32(V, C, like=dy)
        N.call("v100_embedding_bwd", idx, dy, dtable, B, V, C, T)
        return None, dtable


def embedding_bct(idx, table):
    return EmbeddingBCTFn.apply(idx, table)


def _pad_rows(x, lpad: int, tx: int, step: int = 1, off: int = 0, n: Optional[int] = None):
    """Zero-padded copy [B, C, tx] of x[..., off::step][:n] starting at column lpad (v100_pad_copy): the X operand of
    the tap-addressed GEMMs.  64 floats of slack behind it: their 16-byte loads may run a few elements past the last row."""
    B, C, T = x.shape
    n = T if n is None else n
    flat = torch.empty(B * C * tx + 64, dtype=torch.float32, device=x.device)
    xp = flat[:B * C * tx].view(B, C, tx)
    N.call("v100_pad_copy", x, xp, B, C, T, step, off, n, tx, lpad)
    return xp


USE_TAP_GEMM = True      # False forces the explicit im2col / tap-stacking copies (A/B measurements and the equivalence test)


def _taps_ok(b, m, cx, ntap, t, tx, fmt) -> bool:
    return USE_TAP_GEMM and bool(N.helper("v100_pw_taps_supported", b, m, cx, ntap, t, tx, int(fmt)))


def _gemm_taps(W, xp, y, bias, r, m, cx, t, shifts, fmt):
    B, _, tx = xp.shape
    N.call("v100_pw_gemm_taps", W.w, W.w_bf, xp, y, bias, r, B, m, cx, t, tx, len(shifts), (ctypes.c_int * len(shifts))(*shifts), int(fmt))


def _wgrad_taps(g, tg, g_off, xp, m, cx, t, shifts, fmt):
    B, _, tx = xp.shape
    kk = len(shifts) * cx
    S = N.helper("v100_pw_wgrad_splits", B, m, kk)
    partial, dW = _f32(S, m, kk, like=xp), _f32(m, kk, like=xp)
    N.call("v100_pw_wgrad_taps", g, tg, g_off, xp, partial, dW, S, B, m, cx, t, tx, len(shifts), (ctypes.c_int * len(shifts))(*shifts), int(fmt))
    return dW


class ConvTranspose1dK5S2Fn(torch.autograd.Function):
    """nn.ConvTranspose1d(Cin, Cout, kernel_size=5, stride=2, padding=2, bias) -- tts.py:22.

    Output length 2L-1.  Even outputs y[2u] use taps 0/2/4 on x[u+1], x[u], x[u-1]; odd outputs y[2u+1]
    use taps 1/3 on x[u+1], x[u]: two dense GEMMs (K = 3*Cin and 2*Cin) over tap-stacked copies of x, run
    on the pointwise MFMA kernel, then interleaved (+bias).
    """

    @staticmethod
    def _stack(x):
        B, cin, L = x.shape
        xe = _f32(B, 3 * cin, L, like=x)
        xo = _f32(B, 2 * cin, L, like=x)
        for tap, d in enumerate((1, 0, -1)):
            N.call("v100_shift_copy", x, xe, None, B, cin, L, L, cin, 0, 3 * cin, tap * cin, 1, d, 1, 0, L, 0)
        for tap, d in enumerate((1, 0)):
            N.call("v100_shift_copy", x, xo, None, B, cin, L, L, cin, 0, 2 * cin, tap * cin, 1, d, 1, 0, L, 0)
        return xe, xo

    @staticmethod
    def _mats(w):
        # w [Cin][Cout][5] -> Ae [Cout][3*Cin] (taps 0,2,4), Ao [Cout][2*Cin] (taps 1,3); tiny, layout only
        wt = w.detach().permute(1, 2, 0)                      # [Cout][5][Cin]
        ae = wt[:, [0, 2, 4], :].reshape(w.shape[1], -1).contiguous()
        ao = wt[:, [1, 3], :].reshape(w.shape[1], -1).contiguous()
        return ae, ao

    @staticmethod
    def forward(ctx, x, w, bias, precision):
        _check(x, "conv_transpose1d")
        if w.shape[2] != 5:
            raise RuntimeError("conv_transpose1d: only kernel_size=5, stride=2, padding=2 is built (tts.py:22)")
        x = x.contiguous()
        B, cin, L = x.shape
        cout = w.shape[1]
        bf16 = _fmt(precision)
        def _phase_weights():
            ae, ao = ConvTranspose1dK5S2Fn._mats(w)
            return _Weights(ae, bf16, False), _Weights(ao, bf16, False)
        We, Wo = _derived(w, ("convt_fwd", int(bf16)), _phase_weights)
        ye, yo = _f32(B, cout, L, like=x), _f32(B, cout, L, like=x)
        tx = (L + 2 + 3) // 4 * 4
        # no tap-stacked copies: both phases read one zero-padded copy of x through the tap-addressed GEMM
        # (x[u+1], x[u], x[u-1] = xp[u+2], xp[u+1], xp[u]); forward and backward-data shapes must both fit
        ctx.taps = _taps_ok(B, cout, cin, 3, L, tx, bf16) and (bf16 == 2 or _taps_ok(B, cin, cout, 3, L, tx, bf16))
        if ctx.taps:
            xp = _pad_rows(x, 1, tx)
            _gemm_taps(We, xp, ye, None, None, cout, cin, L, (2, 1, 0), bf16)
            _gemm_taps(Wo, xp, yo, None, None, cout, cin, L, (2, 1), bf16)
        else:
            xe, xo = ConvTranspose1dK5S2Fn._stack(x)
            _pw_gemm(We.w, We.w_bf, xe, ye, cout, 3 * cin, L, B, bf16)
            _pw_gemm(Wo.w, Wo.w_bf, xo, yo, cout, 2 * cin, L, B, bf16)
        T = 2 * L - 1
        y = _f32(B, cout, T, like=x)
        b = bias.detach() if bias is not None else None
        N.call("v100_shift_copy", ye, y, b, B, cout, L, T, cout, 0, cout, 0, 1, 0, 2, 0, L, 0)
        if L > 1:
            N.call("v100_shift_copy", yo, y, b, B, cout, L, T, cout, 0, cout, 0, 1, 0, 2, 1, L - 1, 0)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bf16 = bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, cin, L = x.shape
        cout = w.shape[1]
        T = 2 * L - 1
        bf16 = ctx.bf16
        _no_fp16_training(bf16, "backward")
        if ctx.taps:
            return ConvTranspose1dK5S2Fn._backward_taps(ctx, x, w, dy)
        dye, dyo = _f32(B, cout, L, like=x), torch.zeros((B, cout, L), dtype=torch.float32, device=x.device)
        N.call("v100_shift_copy", dy, dye, None, B, cout, T, L, cout, 0, cout, 0, 2, 0, 1, 0, L, 0)
        if L > 1:
            N.call("v100_shift_copy", dy, dyo, None, B, cout, T, L, cout, 0, cout, 0, 2, 1, 1, 0, L - 1, 0)
        xe, xo = ConvTranspose1dK5S2Fn._stack(x)
        ae, ao = ConvTranspose1dK5S2Fn._mats(w)
        We, Wo = _Weights(ae, bf16, True), _Weights(ao, bf16, True)
        # weight gradients: dAe = dye . xe^T, dAo = dyo . xo^T, scattered back to [Cin][Cout][5]
        dae, dao = _f32(cout, 3 * cin, like=x), _f32(cout, 2 * cin, like=x)
        for g, xs, out, kk in ((dye, xe, dae, 3 * cin), (dyo, xo, dao, 2 * cin)):
            S = N.helper("v100_pw_wgrad_splits", B, cout, kk)
            partial = _f32(S, cout, kk, like=x)
            N.call("v100_pw_wgrad", g, None, None, None, None, 0, xs, None, None, 0, partial, out, S, B, cout, kk, L, int(bf16))
        dw = torch.empty_like(w)
        dw[:, :, [0, 2, 4]] = dae.view(cout, 3, cin).permute(2, 0, 1)
        dw[:, :, [1, 3]] = dao.view(cout, 2, cin).permute(2, 0, 1)
        db = None
        if ctx.has_bias:
            G = N.helper("v100_dw_num_groups", B, cout)
            part = _f32(G, cout, 2, like=x)
            N.call("v100_chan_reduce2", dy, None, part, G, B, cout, T)
            db = _f32(cout, like=x)
            N.call("v100_slab_sum0", part, G, db, cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dxe, dxo = _f32(B, 3 * cin, L, like=x), _f32(B, 2 * cin, L, like=x)
            _pw_gemm(We.wt, We.wt_bf, dye, dxe, 3 * cin, cout, L, B, bf16)
            _pw_gemm(Wo.wt, Wo.wt_bf, dyo, dxo, 2 * cin, cout, L, B, bf16)
            dx = torch.zeros_like(x)
            for tap, d in enumerate((1, 0, -1)):        # xe tap block held x[u+d]  ->  dx[t] += dxe[tap][t-d]
                N.call("v100_shift_copy", dxe, dx, None, B, cin, L, L, 3 * cin, tap * cin, cin, 0, 1, -d, 1, 0, L, 1)
            for tap, d in enumerate((1, 0)):
                N.call("v100_shift_copy", dxo, dx, None, B, cin, L, L, 2 * cin, tap * cin, cin, 0, 1, -d, 1, 0, L, 1)
        return dx, dw, db, None


def _convt_backward_taps(ctx, x, w, dy):
    """Backward of ConvTranspose1dK5S2Fn on the tap-addressed GEMMs: dy is split into its even / odd phases straight into
    zero-padded rows (one pass each), which serve as the G operand of the two weight-gradient GEMMs and as the X
    operand of the two data-gradient GEMMs (the second accumulates onto the first: R epilogue)."""
    B, cin, L = x.shape
    cout = w.shape[1]
    T = 2 * L - 1
    bf16 = ctx.bf16
    tx = (L + 2 + 3) // 4 * 4
    xp = _pad_rows(x, 1, tx)
    dyep = _pad_rows(dy, 1, tx, step=2, off=0, n=L)
    dyop = _pad_rows(dy, 1, tx, step=2, off=1 if L > 1 else 0, n=L - 1)
    dae = _wgrad_taps(dyep, tx, 1, xp, cout, cin, L, (2, 1, 0), bf16)
    dao = _wgrad_taps(dyop, tx, 1, xp, cout, cin, L, (2, 1), bf16)
    dw = torch.empty_like(w)
    dw[:, :, [0, 2, 4]] = dae.view(cout, 3, cin).permute(2, 0, 1)
    dw[:, :, [1, 3]] = dao.view(cout, 2, cin).permute(2, 0, 1)
    db = None
    if ctx.has_bias:
        G = N.helper("v100_dw_num_groups", B, cout)
        part = _f32(G, cout, 2, like=x)
        N.call("v100_chan_reduce2", dy, None, part, G, B, cout, T)
        db = _f32(cout, like=x)
        N.call("v100_slab_sum0", part, G, db, cout)
    dx = None
    if ctx.needs_input_grad[0]:
        wd = w.detach()
        # dx[u] = w0 dye[u-1] + w2 dye[u] + w4 dye[u+1]  +  w1 dyo[u-1] + w3 dyo[u];  rows [c][tap*Cout + m]
        a1 = wd[:, :, [0, 2, 4]].permute(0, 2, 1).reshape(cin, 3 * cout).contiguous()
        a2 = wd[:, :, [1, 3]].permute(0, 2, 1).reshape(cin, 2 * cout).contiguous()
        W1, W2 = _Weights(a1, bf16, False), _Weights(a2, bf16, False)
        dx1, dx = _f32(B, cin, L, like=x), _f32(B, cin, L, like=x)
        _gemm_taps(W1, dyep, dx1, None, None, cin, cout, L, (0, 1, 2), bf16)
        _gemm_taps(W2, dyop, dx, None, dx1, cin, cout, L, (0, 1), bf16)
    return dx, dw, db, None


ConvTranspose1dK5S2Fn._backward_taps = staticmethod(_convt_backward_taps)


def conv_transpose1d_k5s2(x, w, bias=None, precision: Optional[str] = None):
    return ConvTranspose1dK5S2Fn.apply(x, w, bias, precision or _PRECISION)


class Conv1dDenseFn(torch.autograd.Function):
    """nn.Conv1d(Cin, Cout, k, stride, padding, bias) with groups=1 -- the dense k=5 convolutions of the v2 conv
    blocks (voice100/models/_layers_v2.py:41-48; config/asr_en_base.yaml:16-18).

    One GEMM on the pointwise MFMA kernel with K = k*Cin over an im2col copy of x (rows tap-major), so every
    prologue / precision mode of K1 applies; backward-weight is the K1 NT kernel against the same copy and
    backward-data is the transposed GEMM followed by col2im (a gather, no atomics).
    """

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, precision):
        _check(x, "conv1d_dense")
        x = x.contiguous()
        B, cin, T = x.shape
        cout, cin_w, k = w.shape
        if cin_w != cin:
            raise RuntimeError(f"conv1d_dense: weight expects {cin_w} input channels, got {cin}")
        tout = (T + 2 * padding - k) // stride + 1
        if tout <= 0:
            raise RuntimeError("conv1d_dense: input shorter than the kernel")
        bf16 = _fmt(precision)
        W = _derived(w, ("dense_fwd", int(bf16)),
                     lambda: _Weights(w.detach().permute(0, 2, 1).reshape(cout, k * cin).contiguous(), bf16, False))    # [Cout][j*Cin + c]
        y = _f32(B, cout, tout, like=x)
        bias_d = bias.detach() if bias is not None else None
        # "same" stride-1 convolutions skip the im2col copy: the GEMM reads a zero-padded copy of x once per tap
        # (tap-addressed X operand, include/voice100_hip.h); forward and backward-data shapes must both fit
        tx = (T + k - 1 + 3) // 4 * 4
        taps = (stride == 1 and 2 * padding == k - 1 and k <= 8 and _taps_ok(B, cout, cin, k, T, tx, bf16)
                and (bf16 == 2 or _taps_ok(B, cin, cout, k, T, tx, bf16)))
        if taps:
            _gemm_taps(W, _pad_rows(x, padding, tx), y, bias_d, None, cout, cin, T, tuple(range(k)), bf16)
        else:
            cols = _f32(B, k * cin, tout, like=x)
            N.call("v100_im2col", x, cols, B, cin, T, tout, k, stride, padding)
            _pw_gemm(W.w, W.w_bf, cols, y, cout, k * cin, tout, B, bf16, bias=bias_d, epi=0)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, bias is not None, bf16, tout, taps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, has_bias, bf16, tout, taps = ctx.cfg
        _no_fp16_training(bf16, "backward")
        dy = dy.contiguous()
        B, cin, T = x.shape
        cout, _, k = w.shape
        kk = k * cin
        tx = (T + k - 1 + 3) // 4 * 4
        if taps:
            dW = _wgrad_taps(dy, T, 0, _pad_rows(x, padding, tx), cout, cin, T, tuple(range(k)), bf16)
        else:
            cols = _f32(B, kk, tout, like=x)
            N.call("v100_im2col", x, cols, B, cin, T, tout, k, stride, padding)
            S = N.helper("v100_pw_wgrad_splits", B, cout, kk)
            partial = _f32(S, cout, kk, like=x)
            dW = _f32(cout, kk, like=x)
            N.call("v100_pw_wgrad", dy, None, None, None, None, 0, cols, None, None, 0, partial, dW, S, B, cout, kk, tout, int(bf16))
        dw = dW.view(cout, k, cin).permute(0, 2, 1).contiguous()
        db = None
        if has_bias:
            G = N.helper("v100_dw_num_groups", B, cout)
            part = _f32(G, cout, 2, like=x)
            N.call("v100_chan_reduce2", dy, None, part, G, B, cout, tout)
            db = _f32(cout, like=x)
            N.call("v100_slab_sum0", part, G, db, cout)
        dx = None
        if ctx.needs_input_grad[0] and taps:
            # dx[c][u] = sum_{j,m} W[m][c][j] dy[m][u + pad - j]: the same GEMM on pad(dy) with the taps reversed
            a = w.detach().flip(2).permute(1, 2, 0).reshape(cin, k * cout).contiguous()       # [Cin][i*Cout + m] = W[m][c][k-1-i]
            dx = _f32(B, cin, T, like=x)
            _gemm_taps(_Weights(a, bf16, False), _pad_rows(dy, padding, tx), dx, None, None, cin, cout, T, tuple(range(k)), bf16)
        elif ctx.needs_input_grad[0]:
            w2d = w.detach().permute(0, 2, 1).reshape(cout, kk).contiguous()
            W = _Weights(w2d, bf16, True)
            dcols = cols                                   # the im2col copy is dead after the wgrad: reuse its storage
            _pw_gemm(W.wt, W.wt_bf, dy, dcols, kk, cout, tout, B, bf16, epi=0)
            dx = _f32(B, cin, T, like=x)
            N.call("v100_col2im", dcols, dx, B, cin, T, tout, k, stride, padding)
        return dx, dw, db, None, None, None


def conv1d_dense(x, w, bias=None, stride: int = 1, padding: int = 0, precision: Optional[str] = None):
    return Conv1dDenseFn.apply(x, w, bias, int(stride), int(padding), precision or _PRECISION)


class LayerNormGeluFn(torch.autograd.Function):
    """gelu(layer_norm over the channel axis) of a [B, C, T] tensor: the `transpose -> nn.LayerNorm(C) -> transpose
    -> F.gelu` tail of the v2 conv blocks (_layers_v2.py:50-56, 83-89) without the transposes."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        _check(y, "layer_norm_gelu")
        y = y.contiguous()
        B, C, T = y.shape
        out = _f32(B, C, T, like=y)
        mean, rstd = _f32(B, T, like=y), _f32(B, T, like=y)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        N.call("v100_ln_gelu_fwd", y, g, b, float(eps), out, mean, rstd, B, C, T)
        ctx.save_for_backward(y, g, b, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, g, b, mean, rstd = ctx.saved_tensors
        dout = dout.contiguous()
        B, C, T = y.shape
        parts = N.helper("v100_ln_num_parts", B, T)
        partial = _f32(parts, C, 2, like=y)
        dy = _f32(B, C, T, like=y)
        N.call("v100_ln_gelu_bwd", dout, y, g, b, mean, rstd, dy, partial, B, C, T)
        dg, db = _f32(C, like=y), _f32(C, like=y)
        N.call("v100_slab_sum2", partial, parts, dg, db, C)
        return dy, dg, db, None


def layer_norm_gelu(y, gamma, beta, eps: float = 1e-5):
    return LayerNormGeluFn.apply(y, gamma, beta, eps)


def world_unnormalize_gate(x_bta, f0_mean, f0_std, ls_mean, ls_std, ca_mean, ca_std):
    """x [B,T,2+S+Cap] -> (f0 [B,T], logspc [B,T,S], codeap [B,T,Cap]); tts.py:192-201."""
    _check(x_bta, "world_unnormalize")
    x_bta = x_bta.contiguous()
    B, T, A = x_bta.shape
    S, cap = ls_mean.shape[0], ca_mean.shape[0]
    assert A == 2 + S + cap
    f0 = _f32(B, T, like=x_bta)
    logspc = _f32(B, T, S, like=x_bta)
    codeap = _f32(B, T, cap, like=x_bta)
    N.call("v100_world_unnormalize", x_bta, f0, logspc, codeap, f0_mean, f0_std, ls_mean, ls_std, ca_mean, ca_std, B, T, S, cap)
    return f0, logspc, codeap



class WorldLossFn(torch.autograd.Function):
    """WORLDLoss (voice100/models/_layers_v1.py:60-93) on the decoder output pred [B, Tp, 2+S+Cap], one HIP pass for the
    four loss terms AND their gradient (v100_world_loss); backward is one scaling kernel.  With `norm` (the six WORLDNorm
    vectors) the targets are the RAW WORLD features and hasf0 = f0 >= 30 is formed in the kernel (tts.py:203-206)."""

    @staticmethod
    def forward(ctx, pred, length, hasf0, f0, logspc, codeap, norm, weights, l1):
        _check(pred, "world_loss")
        pred = pred.contiguous()
        B, Tp, A = pred.shape
        S, cap = logspc.shape[2], codeap.shape[2]
        if A != 2 + S + cap:
            raise RuntimeError(f"world_loss: pred has {A} features, targets imply {2 + S + cap}")
        Tt = f0.shape[1]
        dev = pred.device
        f0, logspc, codeap = (t.to(device=dev, dtype=torch.float32).contiguous() for t in (f0, logspc, codeap))
        if logspc.shape[1] != Tt or codeap.shape[1] != Tt:
            raise RuntimeError("world_loss: targets must share one time axis")
        hasf0 = None if hasf0 is None else hasf0.to(device=dev, dtype=torch.float32).contiguous()
        length = length.to(device=dev, dtype=torch.int32).contiguous()
        nrm = [None] * 6 if norm is None else [t.detach().to(device=dev, dtype=torch.float32).contiguous() for t in norm]
        w = None if weights is None else weights.detach().to(device=dev, dtype=torch.float32).contiguous()
        partial = _f32(N.helper("v100_world_loss_parts", B, Tp), 4, like=pred)
        loss = _f32(4, like=pred)
        unit = torch.empty_like(pred)
        N.call("v100_world_loss", pred, f0, hasf0, logspc, codeap, length, *nrm, w, partial, loss, unit, B, Tp, Tt, S, cap, int(l1))
        ctx.save_for_backward(unit)
        ctx.dims = (B, Tp, S, cap)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (unit,) = ctx.saved_tensors
        B, Tp, S, cap = ctx.dims
        dpred = torch.empty_like(unit)
        N.call("v100_world_loss_bwd", unit, gout.to(torch.float32).contiguous(), dpred, B, Tp, S, cap)
        return (dpred,) + (None,) * 8


def world_loss(pred_bta, length, hasf0, f0, logspc, codeap, norm=None, weights=None, loss: str = "mse"):
    """The four WORLDLoss terms (hasf0, f0, logspc, codeap) as a [4] tensor."""
    if loss not in ("l1", "mse"):
        raise ValueError("Unknown loss type")
    return WorldLossFn.apply(pred_bta, length, hasf0, f0, logspc, codeap, norm, weights, loss == "l1")


class CTCLossFn(torch.autograd.Function):
    """log_softmax(dim=-1) + CTCLoss(blank=0, reduction='mean', zero_infinity=True) on logits [B, T, V]
    (asr.py:146-152), forward and gradient in one pass of the HIP lattice kernels.  Limits of the kernel: V <= 128 classes,
    padded target width <= 2047 tokens (nn.CTCLoss has none; the reference's longest transcripts are a few hundred
    characters).  Labels outside [0, V) count as blank unless VOICE100_CHECK_IDS=1 (then IndexError, like the reference)."""

    @staticmethod
    def forward(ctx, logits, targets, input_lengths, target_lengths, blank, grad_bvt=False):
        _check(logits, "ctc_loss")
        logits = logits.contiguous()
        B, T, V = logits.shape
        targets = targets.to(device=logits.device, dtype=torch.int64).contiguous()
        lmax = targets.shape[1]
        if lmax > 2047:
            raise RuntimeError(f"ctc_loss: padded target width {lmax} exceeds the kernel's 2047-token limit")
        _check_ids(targets, V, "ctc_loss targets")
        il = input_lengths.to(device=logits.device, dtype=torch.int32).contiguous()
        tl = target_lengths.to(device=logits.device, dtype=torch.int32).contiguous()
        nws = N.helper("v100_ctc_workspace_floats", B, T, lmax)
        if nws < 0:
            raise RuntimeError("ctc_loss: workspace too large")
        ws = _f32(nws, like=logits)
        nll = _f32(B, like=logits)
        loss = _f32(1, like=logits)
        # grad_bvt: the gradient buffer is [B, V, T] -- the logits are the transpose of a [B, V, T] tensor (asr.py:114) and that
        # transpose's backward then returns this buffer as it is (TransposeLast2Fn.backward)
        grad = _f32(B, V, T, like=logits) if grad_bvt else torch.empty_like(logits)
        # the 'mean' reduction (finite utterances, / target length, / B) and its factor on the gradient happen in the library
        N.call("v100_ctc_loss_mean_t", logits, targets, il, tl, ws, nll, loss, grad, int(bool(grad_bvt)), B, T, V, lmax, int(blank))
        ctx.save_for_backward(grad)
        ctx.handed_out = False
        ctx.grad_bvt = bool(grad_bvt)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        if ctx.handed_out:
            # the unit-gradient shortcut below gave the SAVED buffer itself downstream (it may since have become a leaf's .grad or been
            # written in place): a second backward through this node (retain_graph=True) can no longer trust it
            raise RuntimeError("ctc_loss: second backward after the unit-root-gradient shortcut handed out the saved gradient; "
                               "call backward() with an explicit gradient tensor (not functional.unit_grad) when retaining the graph")
        if ctx.grad_bvt:
            grad = grad.transpose(1, 2)      # [B, T, V] as autograd expects, strided over the [B, V, T] buffer
        if is_unit_grad(gout):           # the root gradient TrainStep hands to backward(): the factor is exactly 1, no pass over grad
            ctx.handed_out = True
            return grad, None, None, None, None, None
        return grad * gout, None, None, None, None, None


_CTC_BVT = os.environ.get("VOICE100_CTC_BVT", "1") != "0"      # A/B switch: 0 = the gradient in the logits' own layout + a transpose launch

_UNIT_GRADS = {}


def unit_grad(like: torch.Tensor) -> torch.Tensor:
    """A cached scalar 1 of `like`'s device and dtype: the root gradient of loss.backward() without the ones_like launch per step
    (TrainStep passes it; CTCLossFn.backward recognises it by address and skips its multiply).  Nothing may write to it."""
    key = (like.device, like.dtype)
    one = _UNIT_GRADS.get(key)
    if one is None:
        one = _UNIT_GRADS[key] = torch.ones((), device=like.device, dtype=like.dtype)
    return one


def is_unit_grad(g: torch.Tensor) -> bool:
    one = _UNIT_GRADS.get((g.device, g.dtype))
    return one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr()


def tag_half_length(lengths: torch.Tensor, half: torch.Tensor) -> None:
    """Attach (lengths + 1) // 2, already computed on the device by the augmentation pass, to the lengths tensor it belongs to;
    half_length() returns it instead of launching the two integer ops -- as long as `lengths` has not been written to since."""
    lengths._v100_half = (half, lengths._version)


def half_length(lengths: torch.Tensor):
    tag = getattr(lengths, "_v100_half", None)
    if tag is not None and tag[1] == lengths._version and tag[0].device == lengths.device:
        # the public output_length() keeps the reference's contract (asr.py:81-82): same device and dtype as its argument
        return tag[0] if tag[0].dtype == lengths.dtype else tag[0].to(lengths.dtype)
    return None


def ctc_loss(logits_btv, targets, input_lengths, target_lengths, blank: int = 0):
    grad_bvt = bool(getattr(logits_btv, "_v100_grad_T", False)) and logits_btv.requires_grad and _CTC_BVT
    return CTCLossFn.apply(logits_btv, targets, input_lengths, target_lengths, blank, grad_bvt)
