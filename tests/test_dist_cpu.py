"""N > 1 path on CPU: two gloo processes, flat-bucket gradient all-reduce must equal the mean of
the per-rank gradients (DDP semantics), with per-rank (unsynchronised) BatchNorm statistics."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from voice100_amd.dist import FlatGradBuckets, shard_batch
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3, padding=1), torch.nn.BatchNorm1d(8), torch.nn.ReLU6(),
                                torch.nn.Conv1d(8, 2, 1))
    x_all = torch.randn(6, 4, 20)
    lo, hi = shard_batch(6, rank, world)
    buckets = FlatGradBuckets(model.parameters(), bucket_bytes=64)       # tiny buckets: several per step
    assert len(buckets.buckets) > 1
    for step in range(2):
        buckets.begin_step()
        loss = model(x_all[lo:hi]).pow(2).mean()
        loss.backward()
        buckets.finish_step()
    grads = [p.grad.clone() for p in model.parameters()]
    # reference: same model, same shard, no exchange
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3, padding=1), torch.nn.BatchNorm1d(8), torch.nn.ReLU6(),
                              torch.nn.Conv1d(8, 2, 1))
    for step in range(2):
        ref.zero_grad()
        ref(x_all[lo:hi]).pow(2).mean().backward()
    local = [p.grad.clone() for p in ref.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    mean = [sum(g[i] for g in gathered) / world for i in range(len(local))]
    ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(grads, mean))
    # BN running stats stay rank-local (reference DDP semantics: no sync_batchnorm)
    rm = [None] * world
    dist.all_gather_object(rm, model[1].running_mean.clone())
    if rank == 0:
        out.put((ok, bool(not torch.allclose(rm[0], rm[1]))))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, bn_local = out.get(timeout=10)
    assert ok and bn_local


def test_trainstep_through_launcher_gloo_world2(tmp_path):
    """The N > 1 launch path of `python bench.py --gpus N` (trainer.launch_ranks -> torch.distributed.run, rendezvous on
    127.0.0.1) driving the real TrainStep: ranks seeded differently must hold identical weights after construction
    (rank 0's) and after three optimiser steps, with several buckets in flight per step."""
    import sys
    from voice100_amd.trainer import launch_ranks
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_worker.py")
    rc = launch_ranks(worker, [str(tmp_path)], 2, timeout=300)
    assert rc == 0
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert r0["nbuckets"] > 1
    assert not all(torch.equal(a, b) for a, b in zip(r0["before"], r1["before"]))       # seeds really differed
    for a, b, c in zip(r0["after_init"], r1["after_init"], r0["before"]):
        assert torch.equal(a, b) and torch.equal(a, c)                                  # broadcast of rank 0's weights
    for a, b in zip(r0["final"], r1["final"]):
        assert torch.equal(a, b)                                                        # same averaged gradients applied
    assert not torch.allclose(r0["bn_mean"], r1["bn_mean"])                             # BatchNorm statistics stay per rank
    assert r0["losses"] != r1["losses"] and abs(r0["lr"] - 0.98e-2) < 1e-9


def test_second_backward_in_one_step_raises():
    """Gradient accumulation is not supported by the bucket state machine: say so instead of racing the all-reduce."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_double_backward_worker, args=(_free_port(), out))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert out.get(timeout=10) == "raised"


def _double_backward_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from voice100_amd.dist import FlatGradBuckets
    lin = torch.nn.Linear(4, 4)
    buckets = FlatGradBuckets(lin.parameters(), bucket_bytes=16, force_exchange=True)
    buckets.begin_step()
    lin(torch.randn(2, 4)).sum().backward()
    try:
        lin(torch.randn(2, 4)).sum().backward()
        out.put("silent")
    except RuntimeError as e:
        out.put("raised" if "one backward" in str(e) else str(e))
    dist.destroy_process_group()


def _arena_claim_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from voice100_amd import functional as F_
    from voice100_amd.dist import FlatGradBuckets
    lin = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    ps = list(lin.parameters())
    old = FlatGradBuckets(ps, bucket_bytes=1 << 20, force_exchange=True)
    new = FlatGradBuckets(ps, bucket_bytes=1 << 20, force_exchange=True)          # a TrainStep recreated for the same model
    res = []
    a = F_._grad_arena_for(ps[:2], 20)
    res.append(a is not None and a.data_ptr() == new.flat.data_ptr())             # the NEWEST arena serves the request ...
    res.append(F_._grad_arena_for(ps[:2], 20) is None)                            # ... once per step
    new.begin_step()
    res.append(F_._grad_arena_for(ps[:2], 20) is not None)                        # re-armed by begin_step
    new.remove_hooks()
    b = F_._grad_arena_for(ps[:2], 20)
    res.append(b is not None and b.data_ptr() == old.flat.data_ptr())             # unregistered: the older one is next
    del old, b
    import gc
    gc.collect()
    res.append(F_._grad_arena_for(ps[:2], 20) is None)                            # weak references: a dropped arena is gone
    out.put(res)
    dist.destroy_process_group()


def test_grad_arena_is_claimed_once_per_step_and_weakly_held():
    """functional._grad_arena_for / FlatGradBuckets.grad_arena (round-4 advice): a slice of the flat exchange buffer is handed to at
    most one producer per step; the registry prefers the most recently created buffer and does not keep dropped ones alive."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_arena_claim_worker, args=(_free_port(), out))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert out.get(timeout=10) == [True, True, True, True, True]


def test_shard_batch_covers_everything():
    from voice100_amd.dist import shard_batch
    for n in (1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_batch(n, r, w) for r in range(w)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))
