"""World size 2 on ONE GPU: the product training step with a real peer (round-5 review, item 4).  RCCL cannot put two ranks on one
device, so the ranks talk gloo (device tensors go through the host) -- the transport differs from an 8-GPU run, everything above it
(segments, gradient arena, bucket state machine, hooks, mean, FusedAdam on flat views, per-rank BatchNorm) is what such a run executes.
No scaling is measured here."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_product_training_step_world2_on_one_gpu(cuda, tmp_path):
    """launch_ranks starts two FRESH rank processes (children of a child: this pytest process, which has touched the GPU, is never
    re-exec'd).  Reference semantics: DDP = gradient MEAN over ranks, BatchNorm statistics NOT synchronised."""
    from voice100_amd.trainer import launch_ranks
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_gpu_worker.py")
    rc = launch_ranks(worker, [str(tmp_path)], 2, timeout=900)
    assert rc == 0
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(2)]
    for tag, tol in (("small", 1e-6), ("full", 1e-6)):
        a, b = r[0][tag], r[1][tag]
        for x in (a, b):
            assert x["in_flat"]                                   # gradients were exchanged in place in the flat buffer
            assert x["local_vs_mean"] > 1e-3                      # ... and the peers' shards really contributed different gradients
            assert x["grad_err"] <= tol, (tag, x["grad_err"])      # exchanged gradient == mean of the independently computed ones
            assert x["nbuckets"] >= 2
        assert torch.equal(a["weights"], b["weights"])            # identical replicas after the optimiser steps
        assert not torch.equal(a["bn_mean"], b["bn_mean"])        # BatchNorm running statistics stay per-rank
        assert a["orders"] == b["orders"]                         # same collectives in the same order on both ranks
        assert all(o == list(range(a["nbuckets"])) for o in a["orders"])
        assert a["losses"] != b["losses"]                         # rank-local metrics (no sync_dist), different shards
    assert r[0]["full"]["nbuckets"] == 3                          # 46.5 MB of fp32 gradients in 16 MB buckets
