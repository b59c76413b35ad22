"""Worker script of tests/test_infer_cpu.py, started by voice100_amd.trainer.launch_ranks (the same
`torch.distributed.run` path `tools/bench_infer.py --gpus N` takes).  A stock-op stand-in model (the product modules have no
CPU path) is run through voice100_amd.infer.scatter_run in both sharding modes, with outputs that are ragged across ranks;
rank 0 writes the gathered results, and what ONE process computes on the whole batch, to <outdir>/infer.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
from torch import nn

from voice100_amd.infer import scatter_run
from voice100_amd.trainer import init_distributed


def build():
    torch.manual_seed(5)                              # same weights on every rank (replicas)
    return nn.Sequential(nn.Conv1d(4, 8, 3, padding=1), nn.ReLU6(), nn.Conv1d(8, 6, 1)).eval()


def make_fn(net):
    @torch.no_grad()
    def fn(x, lens):
        """x [b, 4, T], lens [b] -> (ids [b, Tmax_of_shard] int64 ragged across shards, n [b] int32, score [b, 3] float)"""
        logits = net(x)                               # per-utterance work only: no cross-sample op
        ids = logits.argmax(1)
        tmax = int(lens.max())
        ids = ids[:, :tmax].clone()
        for i, n in enumerate(lens.tolist()):
            ids[i, n:] = 0
        return ids, lens.to(torch.int32), logits[:, :3, 0].contiguous()
    return fn


def main():
    outdir, n_items = sys.argv[1], int(sys.argv[2])
    subgroup = len(sys.argv) > 3 and sys.argv[3] == "subgroup"
    rank, _, world = init_distributed()
    assert dist.is_initialized() and dist.get_world_size() == world
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n_items, 4, 24, generator=g)
    lens = torch.randint(5, 25, (n_items,), generator=g)
    lens[0] = 24                                      # the global maximum sits in rank 0's shard: other shards are narrower
    fn = make_fn(build())
    res = {}
    if subgroup:
        # a sub-group that does NOT contain global rank 0: its rank 0 is global rank 1, which must receive the result
        grp = dist.new_group(ranks=[1, 2])
        if rank in (1, 2):
            out = scatter_run(fn, (x, lens), mode="round_robin", group=grp)
            assert (out is None) == (rank != 1)
            if rank == 1:
                res["round_robin"] = out
                # fixed-size outputs with the shapes given up front: no agreement round
                fix = lambda a, b: (a.sum((1, 2)), b.to(torch.int32))          # noqa: E731
                res["fixed"] = scatter_run(fix, (x, lens), group=grp, out_specs=[((), torch.float32), ((), torch.int32)])
                res["fixed_single"] = list(fix(x, lens))
                res["single"] = list(fn(x, lens))
                torch.save(res, os.path.join(outdir, "infer.pt"))
            else:
                fix = lambda a, b: (a.sum((1, 2)), b.to(torch.int32))          # noqa: E731
                assert scatter_run(fix, (x, lens), group=grp, out_specs=[((), torch.float32), ((), torch.int32)]) is None
        dist.barrier()
        dist.destroy_process_group()
        return
    for mode in ("contiguous", "round_robin"):
        out = scatter_run(fn, (x, lens), mode=mode)
        assert (out is None) == (rank != 0)
        if rank == 0:
            res[mode] = out
    if rank == 0:
        res["single"] = list(fn(x, lens))
        torch.save(res, os.path.join(outdir, "infer.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
