"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no collective call site of its own; under Lightning's DDP it gets a
bucketed gradient all-reduce (mean) once per step, per-rank BatchNorm statistics and
rank-local metrics.  This module reproduces exactly that exchange:

* gradients are exchanged through ONE flat fp32 buffer (parameters in forward order, 16-byte slots) cut into
  buckets walking the parameters in reverse (the order backward produces them); the stack executor writes a run of
  blocks' gradients straight into its slice of that buffer (`grad_arena`), the few remaining tensors are packed with
  one multi-tensor copy; the mean is taken inside the collective (ncclAvg) on RCCL;
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
    gradients that are not already in the flat buffer (see grad_arena) are packed into it with one multi-tensor
    copy and the bucket's all-reduce is launched asynchronously, so the exchange overlaps the rest of backward.  After finish_step() every
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
        # Layout of the flat buffer: parameters in FORWARD order, every slot starting on a 16-byte boundary.  Forward order, because
        # the stack executor writes a run of blocks' parameter gradients as ONE contiguous array in exactly that order (w1 g1 b1 wd g2
        # b2 w3 g3 b3 per block, functional.IRStackTrainFn): handed the matching slice of this buffer (grad_arena) it writes the
        # gradients where the all-reduce reads them and the per-bucket packing copy disappears (on one rank at the full size: 0.12 ms of
        # multi-tensor copies per step).  Aligned slots, because one odd-sized tensor (the 29-float vocabulary bias) otherwise shifts
        # every later gradient off the 16-byte accesses of the fused Adam kernel (84 instead of 58 us).
        self._view = {}
        off = 0
        for p in self.params:
            self._view[p] = (off, off + p.numel())
            off = (off + p.numel() + 3) & ~3
        self.flat = torch.zeros(off, dtype=dt, device=dev) if self.exchange else None
        # Buckets are cut walking the parameters in REVERSE (the order backward produces gradients) and launched strictly in that
        # index order; each is one contiguous range of the flat buffer.
        self._bucket_of = {}
        self.buckets = []          # (start, end, [params])
        itemsize = 4 if self.flat is None else self.flat.element_size()
        members, hi = [], None
        for p in reversed(self.params):
            s0, e0 = self._view[p]
            if hi is None:
                hi = (e0 + 3) & ~3
            self._bucket_of[p] = len(self.buckets)
            members.append(p)
            if (hi - s0) * itemsize >= bucket_bytes:
                self.buckets.append((s0, hi, members))
                members, hi = [], None
        if members:
            self.buckets.append((self._view[members[-1]][0], hi, members))
        # mean = sum / world: RCCL averages inside the collective (ncclAvg); gloo (the CPU tests) has no AVG: divide afterwards
        # (a group of ONE -- force_exchange, bench.py's dp_path_single_rank -- sums: RCCL implements a one-rank AVG as a separate
        # pre-multiply pass over the buffer, 39 us per 16 MB bucket, which no rank of a real job runs)
        self._avg_in_collective = self.exchange and self.world > 1 and dist.get_backend(process_group) == "nccl"
        if self._avg_in_collective:
            # probe once (every rank takes part): a backend build without ncclAvg falls back to sum + divide instead of failing mid-step
            try:
                probe = torch.ones(4, dtype=dt, device=dev)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                if not bool((probe == 1).all()):
                    self._avg_in_collective = False
            except RuntimeError as e:
                # only "this backend build has no AVG" is a reason to fall back; any other failure of a collective (a dead peer,
                # a communicator in an error state) must surface here, not as a hang or a wrong sum in the first step
                msg = str(e).lower()
                if not ("avg" in msg or "not supported" in msg or "unsupported" in msg or "invalid argument" in msg):
                    raise
                self._avg_in_collective = False
        self._pending = [0] * len(self.buckets)
        self._next = 0             # buckets are launched strictly in index order, so every rank issues the same collectives
        self._handles = []
        self._hooks = []
        if self.exchange:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            from . import functional as F_
            F_.register_grad_arena(self)
        self.begin_step()

    def grad_arena(self, tensors):
        """The slice of the flat buffer in which `tensors` (parameters, in order) have their gradient slots back to back, or None when
        they are not consecutive / padded apart / not ours: a producer that writes its gradients as one array writes them HERE."""
        if not self.exchange or not tensors:
            return None
        first = self._view.get(tensors[0])
        if first is None:
            return None
        pos = first[0]
        for t in tensors:
            v = self._view.get(t)
            if v is None or v[0] != pos:
                return None
            pos = v[1]
            if self._pending[self._bucket_of[t]] < 0:
                return None                                # its bucket is already on the wire this step
        # A slice is handed out at most ONCE per step.  Two backward nodes of the same stack in one graph (two forwards, one backward;
        # shared weights) both run before AccumulateGrad has set w.grad, so both would be handed this slice, the second would
        # overwrite the first and autograd would sum two aliases of it (G2 + G2 instead of G1 + G2): the second node gets None
        # and takes the ordinary path (its own buffer; autograd accumulates, _launch packs the sum).
        if first[0] in self._claimed:
            return None
        self._claimed.add(first[0])
        return self.flat[first[0]:pos]

    def begin_step(self):
        """Drop last step's gradients (autograd then assigns instead of accumulating) and re-arm the buckets."""
        for p in self.params:
            p.grad = None
        self._claimed = set()
        for i, (_, _, members) in enumerate(self.buckets):
            self._pending[i] = len(members)
        self._next = 0
        self._handles = []
        self.launch_order = []     # bucket indices in the order this step put them on the wire (every rank must show the same list)

    def _launch(self, b):
        s, e, members = self.buckets[b]
        self.launch_order.append(b)
        views, grads = [], []
        for p in members:
            v = self.flat[self._view[p][0]:self._view[p][1]].view_as(p)
            g = p.grad
            if g is not None and g.data_ptr() == v.data_ptr():
                continue                                   # written in place by its producer (grad_arena): nothing to pack
            views.append(v)
            grads.append(g if g is not None else torch.zeros_like(p))
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self._handles.append(dist.all_reduce(self.flat[s:e], op=op, group=self.group, async_op=True))
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
        if not self._avg_in_collective and self.world > 1:
            self.flat.div_(self.world)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import functional as F_
        F_.unregister_grad_arena(self)


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
