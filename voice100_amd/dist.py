"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no collective call site of its own; under Lightning's DDP it gets a
bucketed gradient all-reduce (mean) once per step, per-rank BatchNorm statistics and
rank-local metrics.  This module reproduces exactly that exchange:

* all gradients live in ONE flat fp32 buffer (p.grad are views into it), cut into
  buckets in reverse parameter order (the order backward produces them);
* a post-accumulate hook per parameter launches the bucket's all-reduce as soon as its
  last gradient is written, so the exchange overlaps the rest of backward;
* xGMI is point-to-point (7 links/GPU), the whole message is 46.5 MB, so a few large
  buckets beat many small ones: default 16 MB.

Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" for the CPU tests).
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradBuckets:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("parameters must share one device and dtype")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        # reverse order: the last layers' gradients are ready first
        order = list(reversed(self.params))
        self._bucket_of = {}
        self.buckets = []          # (start, end, n_params)
        off, start, count = 0, 0, 0
        itemsize = self.flat.element_size()
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            count += 1
            if (off - start) * itemsize >= bucket_bytes:
                self.buckets.append((start, off, count))
                start, count = off, 0
        if count:
            self.buckets.append((start, off, count))
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.begin_step()

    def begin_step(self):
        """Zero the flat buffer (one memset) and re-arm the bucket counters; call before backward."""
        self.flat.zero_()
        for i, (_, _, n) in enumerate(self.buckets):
            self._pending[i] = n
        self._handles = []

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if p.grad.data_ptr() != self.flat.data_ptr() + self._offset_bytes(p):
            # autograd replaced the view (first backward with a None grad cannot happen: grads are preset)
            raise RuntimeError("parameter gradient no longer aliases the flat buffer")
        self._pending[b] -= 1
        if self._pending[b] == 0:
            s, e, _ = self.buckets[b]
            self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _offset_bytes(self, p):
        return p.grad.storage_offset() * self.flat.element_size()

    def finish_step(self):
        """Wait for the exchange and turn the sum into DDP's mean.  Call after backward."""
        if self.world == 1:
            return
        if any(n != 0 for n in self._pending):      # a parameter produced no gradient this step
            for h in self._handles:
                h.wait()
            self._handles = [dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
            # buckets already reduced would be summed twice: rescale them
            for b, (s, e, _) in enumerate(self.buckets):
                if self._pending[b] == 0:
                    self.flat[s:e].div_(self.world)
        for h in self._handles:
            h.wait()
        self.flat.div_(self.world)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of a global batch for this rank (pure data parallelism; utterances are
    independent)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
