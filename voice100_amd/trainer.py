"""Minimal step loop replacing `pytorch_lightning.Trainer.fit` for the v1 training scripts
(voice100/train_asr.py:12-38): Adam (L2-style weight_decay as torch.optim.Adam), StepLR(0.98) per
epoch, one gradient all-reduce per step when launched with one process per GPU."""
import os
from typing import Optional

import torch
import torch.distributed as dist

from .dist import FlatGradBuckets


def init_distributed(backend: Optional[str] = None):
    """(rank, local_rank, world) from the torchrun environment; initialises the process group if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return rank, local_rank, world


class TrainStep:
    """One optimisation step of a module exposing training_step(batch, idx) and configure_optimizers()."""

    def __init__(self, model: torch.nn.Module, bucket_bytes: int = 16 << 20):
        self.model = model
        cfg = model.configure_optimizers()
        if isinstance(cfg, dict):
            self.optimizer, self.scheduler = cfg["optimizer"], cfg.get("lr_scheduler")
        else:
            self.optimizer, self.scheduler = cfg, None
        self.buckets = FlatGradBuckets(model.parameters(), bucket_bytes)
        self.step_idx = 0

    def __call__(self, batch) -> torch.Tensor:
        self.model.train()
        self.buckets.begin_step()
        loss = self.model.training_step(batch, self.step_idx)
        if isinstance(loss, dict):
            loss = loss["loss"]
        loss.backward()
        self.buckets.finish_step()
        self.optimizer.step()
        self.step_idx += 1
        return loss.detach()

    def end_epoch(self):
        if self.scheduler is not None:
            self.scheduler.step()
