"""Worker of tests/test_gpu_dist2.py, started by voice100_amd.trainer.launch_ranks as TWO ranks that share cuda:0 (the GPU box has one
GPU; RCCL refuses two ranks on one device, gloo moves device tensors through the host): the PRODUCT training step -- AudioToTextCTC on
the HIP stack executor cut into three-block autograd segments, the flat gradient buffer used as the executor's gradient arena, buckets
launched from post-accumulate hooks, FusedAdam on the flat views -- with a real peer.  Each rank checks, for a small model over three
steps and for the full-width model (asr_en_base, bf16 operands + 16-bit activation storage: what bench.py runs) over one step:
  * its exchanged gradient == mean over the ranks of the gradients each rank computes WITHOUT any exchange (twin model, same shard, same seeds);
  * and writes weights / BatchNorm statistics / bucket launch order to <outdir>/rank<r>.pt for the parent to compare across ranks."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def flat_cpu(tensors):
    return torch.cat([t.detach().reshape(-1).float().cpu() for t in tensors])


def run_case(tag, dims, B, T, L, steps, precision, bucket_bytes, rank, world, dev, out):
    from voice100_amd import functional as F_
    from voice100_amd.asr import AudioToTextCTC
    from voice100_amd.trainer import TrainStep
    F_.set_matmul_precision(precision)
    try:
        torch.manual_seed(1000 + rank)                           # deliberately different initial weights and BN buffers per rank
        model = AudioToTextCTC(*dims).to(dev)
        step = TrainStep(model, bucket_bytes=bucket_bytes)       # construction-time broadcast: rank 0's state everywhere
        assert step.buckets.exchange and step.buckets.world == world and len(step.buckets.buckets) >= 2
        twin = AudioToTextCTC(*dims).to(dev)
        twin.load_state_dict(model.state_dict())
        twin.train()
        g = torch.Generator().manual_seed(50 + rank)             # a different shard per rank
        audio = (torch.randn(B, T, 64, generator=g) * 2 - 4).to(dev)
        alen = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32).to(dev)
        text = torch.randint(1, dims[2], (B, L), generator=g).to(dev)
        tlen = torch.randint(max(1, L // 2), L + 1, (B,), generator=g).to(torch.int32).to(dev)
        batch = ((audio, alen), (text, tlen))

        def seed(i):                                             # python `random` draws the augmentation decisions, torch the dropout seed
            random.seed(7 + i); torch.manual_seed(7 + i)

        # the gradients this rank computes on its shard with NO exchange (plain autograd, no buckets on the twin's parameters)
        seed(0)
        twin.training_step(batch, 0).backward()
        local = flat_cpu([p.grad for p in twin.parameters()])
        want = local.clone()
        dist.all_reduce(want)                                    # CPU tensors over gloo: the mean the exchange must reproduce
        want /= world
        orders, losses = [], []
        for i in range(steps):
            seed(i)
            losses.append(float(step(batch)))
            orders.append(list(step.buckets.launch_order))
            if i == 0:
                got = flat_cpu([p.grad for p in model.parameters()])
                # every gradient of the model lives in the flat buffer after the exchange (views), written in place by the stack executor
                lo = step.buckets.flat.data_ptr()
                hi = lo + step.buckets.flat.numel() * 4
                in_flat = all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
                scale = float(want.abs().max())
                err = float((got - want).abs().max()) / max(scale, 1e-30)
                differs = float((local - want).abs().max()) / max(scale, 1e-30)      # the peers' shards really differ
        torch.cuda.synchronize()
        bn = [b_ for n_, b_ in model.named_buffers() if n_.endswith("running_mean")]
        out[tag] = {"grad_err": err, "grad_equal": bool(torch.equal(got, want)), "local_vs_mean": differs, "in_flat": in_flat,
                    "weights": flat_cpu(list(model.parameters())), "bn_mean": flat_cpu(bn), "orders": orders, "losses": losses,
                    "nbuckets": len(step.buckets.buckets), "segment": 3 if world > 1 else None}
        step.buckets.remove_hooks()
    finally:
        F_.set_matmul_precision("fp32")


def main():
    outdir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")                              # BEFORE anything touches the GPU in this fresh process
    assert dist.get_world_size() == world == 2
    dev = torch.device("cuda", 0)                                # both ranks on the one GPU of the box
    torch.cuda.set_device(dev)
    out = {"rank": rank}
    # small model, fp32, three FusedAdam steps, tiny buckets (many collectives in flight per step)
    run_case("small", (64, 32, 29, 32), 4, 96, 10, 3, "fp32", 1 << 14, rank, world, dev, out)
    # asr_en_base at full width, bf16 operands + activation storage level 5 (bench.py's step), the default 16 MB buckets: 46.5 MB flat
    run_case("full", (64, 512, 29, 512), 4, 256, 24, 1, "bf16", 16 << 20, rank, world, dev, out)
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
