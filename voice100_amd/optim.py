"""Adam as the reference's models configure it (voice100/models/asr.py:169-176, tts.py:132-135, 239-241), one kernel launch per
step for the whole model (csrc/adam.hip).  A regular torch.optim.Optimizer -- param_groups, state_dict, LR schedulers
(StepLR in asr.py:175) all work -- whose step() runs on the HIP library when every parameter is a float32 CUDA tensor and
falls back to nothing else: CPU parameters raise (use torch.optim.Adam there; the product path is the GPU)."""
import ctypes

import numpy as np
import torch

from . import _native as N


RING = 32


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = None

    # The device-side tables (flat moment buffers, pointer arrays, step counter) are a cache of self.state / param_groups:
    # anything that replaces those -- load_state_dict() after a step has run (in-place resume, roll-back to a checkpoint),
    # add_param_group() -- drops the cache, and the next step() rebuilds it from self.state (keeping the loaded moments).
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = None

    def _build(self, group):
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("FusedAdam: parameters must be contiguous float32 CUDA tensors on one device")
        n = sum(p.numel() for p in ps)
        flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        ce = N.helper("v100_adam_chunk_elems")
        chunks, off = [], 0
        m_ptrs, v_ptrs, p_ptrs = [], [], []
        for ti, p in enumerate(ps):
            k = p.numel()
            st = self.state[p]
            if "exp_avg" in st:                              # resumed from a state_dict: keep its moments
                flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            st["exp_avg"] = flat_m[off:off + k].view_as(p)
            st["exp_avg_sq"] = flat_v[off:off + k].view_as(p)
            st.setdefault("step", torch.tensor(0.0))
            m_ptrs.append(flat_m.data_ptr() + 4 * off)
            v_ptrs.append(flat_v.data_ptr() + 4 * off)
            p_ptrs.append(p.data_ptr())
            for o in range(0, k, ce):
                chunks.append((ti, min(ce, k - o), o))
            off += k
        rec = np.zeros(len(chunks), dtype=np.dtype([("tensor", "<i4"), ("count", "<i4"), ("offset", "<i8")]))
        for i, (ti, c, o) in enumerate(chunks):
            rec[i] = (ti, c, o)
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).to(dev)
        t = {"params": ps, "chunks": to_dev(rec), "nchunks": len(chunks), "p": to_dev(np.array(p_ptrs, dtype=np.uint64)),
             "m": to_dev(np.array(m_ptrs, dtype=np.uint64)), "v": to_dev(np.array(v_ptrs, dtype=np.uint64)),
             "g": torch.empty(8 * len(ps), dtype=torch.uint8, device=dev), "flat": (flat_m, flat_v), "step": 0,
             "p_ptrs": p_ptrs,
             # pinned staging buffers for the per-step gradient-pointer upload, used round-robin; an event per buffer says when
             # its async copy has executed, and is waited for before the buffer is rewritten RING steps later (a no-op unless
             # the host runs more than RING steps ahead of the GPU: with four slots that wait was 0.55 ms of every step's 2.6 ms of
             # enqueue time in a GPU-bound loop -- back-pressure, not work; 32 slots of 1.3 KB keep it out of the enqueue path)
             "ring": [torch.empty(8 * len(ps), dtype=torch.uint8).pin_memory() for _ in range(RING)], "events": [None] * RING, "pos": 0}
        t["step"] = int(max(float(self.state[p]["step"]) for p in ps))
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._tables is None:
            self._tables = [self._build(g) for g in self.param_groups]
        for group, t in zip(self.param_groups, self._tables):
            if t is None:
                continue
            ps = t["params"]
            grads = [p.grad for p in ps]
            f32 = torch.float32
            try:
                # one pass: a gradient that is not a contiguous fp32 CUDA tensor (None included) takes the slow path below
                bad = [i for i, g in enumerate(grads) if g.dtype is not f32 or not g.is_contiguous() or not g.is_cuda]
            except AttributeError:
                raise RuntimeError("FusedAdam: every parameter needs a gradient each step (the reference's models produce one)") from None
            for i in bad:
                grads[i] = ps[i].grad = grads[i].to(device=ps[i].device, dtype=f32).contiguous()
            ptrs = [g.data_ptr() for g in grads]
            if [p.data_ptr() for p in ps] != t["p_ptrs"]:        # every step: far cheaper than a write into freed memory
                raise RuntimeError("FusedAdam: a parameter's storage moved since the optimizer was built (re-create the optimizer)")
            if ptrs != t.get("g_ptrs"):
                # the gradient tensors moved (the caching allocator usually hands back the same blocks every step): upload
                slot = t["pos"] % RING
                t["pos"] += 1
                host = t["ring"][slot]
                if t["events"][slot] is not None:
                    t["events"][slot].synchronize()
                host.numpy().view(np.uint64)[:] = ptrs
                t["g"].copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                t["events"][slot] = ev
                t["g_ptrs"] = ptrs
            t["step"] += 1
            b1, b2 = group["betas"]
            N.call("v100_adam_step", t["chunks"], t["nchunks"], t["p"], t["g"], t["m"], t["v"], float(group["lr"]), float(b1), float(b2),
                   float(group["eps"]), float(group["weight_decay"]), t["step"])
            # the kernel wrote the parameters through raw pointers: advance their version counters (host-side metadata), or everything
            # keyed on them -- the eval-mode caches of folded coefficients / 16-bit weight copies -- would keep serving the old weights
            torch.autograd.graph.increment_version(ps)
            # state[p]["step"]: ONE host tensor shared by the group's parameters (attached once, advanced in place) -- what torch's
            # Adam keeps per parameter, without 170 dictionary writes a step
            st = t.get("step_tensor")
            if st is None or any(self.state[p].get("step") is not st for p in ps[:1]):
                st = t["step_tensor"] = torch.tensor(float(t["step"]))
                for p in ps:
                    self.state[p]["step"] = st
            else:
                st.fill_(float(t["step"]))
        return loss
