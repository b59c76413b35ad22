"""Minimal step loop replacing `pytorch_lightning.Trainer.fit` for the v1 training scripts
(voice100/train_asr.py:12-38): Adam (L2-style weight_decay as torch.optim.Adam), StepLR(0.98) per
epoch, one gradient all-reduce per step when launched with one process per GPU."""
import os
import signal
import socket
import subprocess
import sys
from typing import List, Optional

import torch
import torch.distributed as dist

from . import functional as F_
from .dist import FlatGradBuckets, broadcast_module_state


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


def launch_ranks(script: str, argv: List[str], nproc: int, timeout: Optional[float] = None) -> int:
    """Start `nproc` ranks of `script` on this node -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> script argv...` -- as a CHILD process and return its exit code.
    Call it before anything in the calling process has touched the GPU (a process that has initialised HIP must not
    fork GPU work; the children are fresh interpreters, nothing is re-exec'd)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    # own session: on a timeout the WHOLE group (torchrun and the rank processes it forked) is ended, so no orphaned rank
    # keeps a GPU; only the group this call created is signalled, never a pattern
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        raise


_PRECISIONS = {"32": "fp32", "32-true": "fp32", "fp32": "fp32",
               "bf16": "bf16", "bf16-mixed": "bf16",
               # Lightning's fp16 AMP recipe (README.md:189, 212, 233, 284: `--trainer.precision 16`).  On MI355X the 16-bit matrix rate and
               # the memory of bf16 and fp16 are the same and bf16 needs no loss scaling: the recipe trains in "bf16" here
               "16": "bf16", "16-mixed": "bf16"}


def resolve_precision(precision) -> str:
    """The `--trainer.precision` values of the reference's training recipes (pytorch_lightning.Trainer: 32, 16, "bf16", and the
    "-true" / "-mixed" spellings of later releases) -> this library's matmul precision ("fp32" | "bf16")."""
    key = str(precision).lower()
    if key not in _PRECISIONS:
        raise ValueError(f"precision must be one of 32, 16, 'bf16' (or '32-true', '16-mixed', 'bf16-mixed'), got {precision!r}")
    return _PRECISIONS[key]


class TrainStep:
    """One optimisation step of a module exposing training_step(batch, idx) and configure_optimizers().
    With more than one rank the constructor first copies rank 0's parameters and buffers to every rank (DDP's
    construction-time broadcast), so replicas cannot start from different weights.
    precision: the reference recipes' `--trainer.precision` (32 | 16 | "bf16"; None = leave functional.set_matmul_precision as it is).
    16 -- Lightning's fp16 autocast + GradScaler -- runs as "bf16" (same rate and memory here, no loss scaling needed)."""

    def __init__(self, model: torch.nn.Module, bucket_bytes: int = 16 << 20, force_exchange: bool = False, precision=None):
        if precision is not None:
            F_.set_matmul_precision(resolve_precision(precision))
        # force_exchange: run the bucketed all-reduce machinery even in a process group of ONE rank (bench.py's
        # `dp_path_single_rank`: what the data-parallel path costs at full size, measurable without a node)
        self.model = model
        broadcast_module_state(model)
        cfg = model.configure_optimizers()
        if isinstance(cfg, dict):
            self.optimizer, self.scheduler = cfg["optimizer"], cfg.get("lr_scheduler")
        else:
            self.optimizer, self.scheduler = cfg, None
        self.buckets = FlatGradBuckets(model.parameters(), bucket_bytes, force_exchange=force_exchange)
        self.step_idx = 0

    def __call__(self, batch) -> torch.Tensor:
        if not self.model.training:           # (train() walks every sub-module: ~0.5 ms of host time per call on asr_en_base)
            self.model.train()
        self.buckets.begin_step()
        loss = self.model.training_step(batch, self.step_idx)
        if isinstance(loss, dict):
            loss = loss["loss"]
        if loss.dim() == 0 and loss.is_cuda:
            torch.autograd.backward(loss, grad_tensors=F_.unit_grad(loss))      # (no ones_like launch; the CTC head sees the factor is 1)
        else:
            loss.backward()
        self.buckets.finish_step()
        self.optimizer.step()
        self.step_idx += 1
        return loss.detach()

    def end_epoch(self):
        if self.scheduler is not None:
            self.scheduler.step()
