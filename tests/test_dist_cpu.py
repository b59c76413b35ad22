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


def test_shard_batch_covers_everything():
    from voice100_amd.dist import shard_batch
    for n in (1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_batch(n, r, w) for r in range(w)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))
