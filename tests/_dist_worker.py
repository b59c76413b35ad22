"""Worker script of tests/test_dist_cpu.py::test_trainstep_through_launcher_gloo_world2, started by
voice100_amd.trainer.launch_ranks (the same `torch.distributed.run` path `python bench.py --gpus N` takes).
Each rank: init_distributed() -> a small module with the LightningModule hooks, seeded DIFFERENTLY per rank ->
TrainStep (construction-time broadcast + bucketed gradient mean + Adam) -> results to <outdir>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
from torch import nn

from voice100_amd._base import Voice100ModelBase
from voice100_amd.dist import shard_batch
from voice100_amd.trainer import TrainStep, init_distributed


class Toy(Voice100ModelBase):
    """Stock-op stand-in with the hooks TrainStep drives (the product modules have no CPU path)."""

    def __init__(self, hidden, learning_rate):
        super().__init__()
        self.save_hyperparameters()
        self.net = nn.Sequential(nn.Conv1d(4, hidden, 3, padding=1), nn.BatchNorm1d(hidden), nn.ReLU6(), nn.Conv1d(hidden, 2, 1))

    def training_step(self, batch, batch_idx=0):
        x, y = batch
        loss = (self.net(x) - y).pow(2).mean()
        self.log("train_loss", loss)
        return loss

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.parameters(), lr=self.hparams.learning_rate)
        return {"optimizer": opt, "lr_scheduler": torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.98)}


def main():
    outdir = sys.argv[1]
    rank, local_rank, world = init_distributed()
    assert dist.is_initialized() and dist.get_world_size() == world == 2
    torch.manual_seed(100 + rank)                       # deliberately different initial weights per rank
    model = Toy(hidden=8, learning_rate=1e-2)
    before = [p.detach().clone() for p in model.parameters()]
    step = TrainStep(model, bucket_bytes=64)            # tiny buckets: several collectives per step
    after_init = [p.detach().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(6, 4, 20, generator=g), torch.randn(6, 2, 20, generator=g)
    lo, hi = shard_batch(6, rank, world)
    losses = [float(step((x_all[lo:hi], y_all[lo:hi]))) for _ in range(3)]
    step.end_epoch()
    torch.save({"before": before, "after_init": after_init, "final": [p.detach().clone() for p in model.parameters()],
                "bn_mean": model.net[1].running_mean.clone(), "losses": losses, "nbuckets": len(step.buckets.buckets),
                "lr": step.optimizer.param_groups[0]["lr"], "logged": float(model.logged_metrics["train_loss"])
                if hasattr(model, "logged_metrics") else None},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
