"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no collective call site of its own; under Lightning's DDP it gets a
bucketed gradient all-reduce (mean) once per step, per-rank BatchNorm statistics and
rank-local metrics.  This module reproduces exactly that exchange:

* gradients are exchanged through ONE flat fp32 buffer cut into buckets in reverse parameter
  order (the order backward produces them); a bucket is packed with one multi-tensor copy;
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
    """Bucketed gradient mean over the process group.

    Parameters are cut into buckets in reverse order (the order backward produces gradients).  A
    post-accumulate hook per parameter counts a bucket down; when its last gradient has been written the
    bucket's gradients are packed into the flat buffer with one multi-tensor copy and its all-reduce is
    launched asynchronously, so the exchange overlaps the rest of backward.  After finish_step() every
    p.grad is a view into the (averaged) flat buffer.  With a single process nothing is registered and
    gradients stay where autograd put them (no extra kernels on the step).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, process_group=None,
                 force_exchange: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("parameters must share one device and dtype")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_exchange: run the bucket / all-reduce machinery even in a group of one (tests of the RCCL path on one GPU)
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev) if self.exchange else None
        order = list(reversed(self.params))
        self._bucket_of = {}
        self.buckets = []          # (start, end, [params])
        off, start, members = 0, 0, []
        itemsize = 4 if self.flat is None else self.flat.element_size()
        self._view = {}
        for p in order:
            n = p.numel()
            self._view[p] = (off, off + n)
            self._bucket_of[p] = len(self.buckets)
            off += n
            members.append(p)
            if (off - start) * itemsize >= bucket_bytes:
                self.buckets.append((start, off, members))
                start, members = off, []
        if members:
            self.buckets.append((start, off, members))
        self._pending = [0] * len(self.buckets)
        self._next = 0             # buckets are launched strictly in index order, so every rank issues the same collectives
        self._handles = []
        self._hooks = []
        if self.exchange:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.begin_step()

    def begin_step(self):
        """Drop last step's gradients (autograd then assigns instead of accumulating) and re-arm the buckets."""
        for p in self.params:
            p.grad = None
        for i, (_, _, members) in enumerate(self.buckets):
            self._pending[i] = len(members)
        self._next = 0
        self._handles = []

    def _launch(self, b):
        s, e, members = self.buckets[b]
        views = [self.flat[self._view[p][0]:self._view[p][1]].view_as(p) for p in members]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in members]
        torch._foreach_copy_(views, grads)
        for p, v in zip(members, views):
            p.grad = v
        self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._pending[b] = -1

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if self._pending[b] <= 0:
            # a second backward inside one begin_step()/finish_step() pair would accumulate into p.grad, which by now is a
            # view of the flat buffer an all-reduce may still be reading
            raise RuntimeError("FlatGradBuckets: a gradient arrived for a bucket that was already exchanged this step; "
                               "exactly one backward per begin_step()/finish_step() (no gradient accumulation)")
        self._pending[b] -= 1
        # launch in index order only: a rank on which some parameter received no gradient must not issue its
        # collectives in a different order (sizes differ per bucket) than the other ranks
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def finish_step(self):
        """Wait for the exchange and turn the sum into DDP's mean.  Call after backward."""
        if not self.exchange:
            return
        while self._next < len(self.buckets):      # buckets with a parameter that produced no gradient this step
            self._launch(self._next)
            self._next += 1
        for h in self._handles:
            h.wait()
        self.flat.div_(self.world)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_module_state(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Copy rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does at construction):
    afterwards all replicas start from identical weights, whatever each rank seeded or loaded.  One flat message per
    dtype; a no-op without an initialised process group or in a group of one."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    tensors = [t.detach() for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dt, dev), group in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in group]) if group else None
        dist.broadcast(flat, src=src, group=process_group)
        off = 0
        with torch.no_grad():
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of a global batch for this rank (pure data parallelism; utterances are
    independent)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
