"""Inference data-parallel scatter (SURVEY.md 8e "Inference", BASELINE configs[4]) on CPU: world-size-2 gloo through the same
launcher `tools/bench_infer.py --gpus N` uses; the gathered output must equal the single-process output bit for bit."""
import os

import pytest
import torch

from voice100_amd.infer import shard_indices, scatter_run


def test_shard_indices_partition():
    for n in (0, 1, 7, 32, 257):
        for w in (1, 2, 3, 8):
            for mode in ("contiguous", "round_robin"):
                parts = [shard_indices(n, r, w, mode) for r in range(w)]
                allidx = torch.cat(parts).sort().values
                assert torch.equal(allidx, torch.arange(n)), (n, w, mode)
                assert max(p.numel() for p in parts) <= (n + w - 1) // w
    assert shard_indices(10, 1, 4, "round_robin").tolist() == [1, 5, 9]
    with pytest.raises(ValueError):
        shard_indices(4, 4, 4)
    with pytest.raises(ValueError):
        shard_indices(4, 0, 2, "zigzag")


def test_scatter_run_single_process_is_identity():
    x = torch.arange(12.).reshape(6, 2)
    out = scatter_run(lambda a: (a * 2, a.sum(1)), (x,))
    assert torch.equal(out[0], x * 2) and torch.equal(out[1], x.sum(1))
    with pytest.raises(ValueError):
        scatter_run(lambda a, b: (a,), (x, x[:3]))


@pytest.mark.parametrize("n_items", [7, 1])          # 7: uneven shards (4 + 3); 1: a rank with no work at all
def test_scatter_gather_gloo_world2_equals_single_process(tmp_path, n_items):
    from voice100_amd.trainer import launch_ranks
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_infer_worker.py")
    rc = launch_ranks(worker, [str(tmp_path), str(n_items)], 2, timeout=300)
    assert rc == 0
    res = torch.load(tmp_path / "infer.pt")
    ids1, n1, sc1 = res["single"]
    for mode in ("contiguous", "round_robin"):
        ids, n, sc = res[mode]
        assert ids.dtype == torch.int64 and n.dtype == torch.int32
        assert torch.equal(n, n1) and torch.equal(sc, sc1)            # bit for bit
        assert ids.shape == ids1.shape and torch.equal(ids, ids1)


def test_scatter_gather_subgroup_without_global_rank0(tmp_path):
    """world 3, group = ranks {1, 2}: the group's rank 0 is global rank 1 (dist.gather's dst is a global rank), and the
    out_specs form (no object all-gather) returns the same values."""
    from voice100_amd.trainer import launch_ranks
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_infer_worker.py")
    rc = launch_ranks(worker, [str(tmp_path), "5", "subgroup"], 3, timeout=300)
    assert rc == 0
    res = torch.load(tmp_path / "infer.pt")
    for got, want in zip(res["round_robin"], res["single"]):
        assert torch.equal(got, want)
    for got, want in zip(res["fixed"], res["fixed_single"]):
        assert got.dtype == want.dtype and torch.equal(got, want)
