"""N > 1 path on CPU: two gloo ranks shard a batch of pairs and rank 0 reassembles the flows in
global order (no data-path collective)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_partition():
    sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
    from src.dist import shard_range
    for n in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_total, out_path):
    sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
    from src.dist import gather_flows, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # stand-in for the per-rank engine output: pair i -> constant field i (+ rank-independent pattern)
    local = torch.stack([torch.full((4, 6, 2), float(i)) for i in range(lo, hi)]) if hi > lo \
        else torch.zeros((0, 4, 6, 2))
    got = gather_flows(local, n_total)
    if rank == 0:
        np.save(out_path, got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 8])
def test_two_rank_shard_and_gather(tmp_path, n_total):
    out = str(tmp_path / "g.npy")
    port = 29500 + (os.getpid() % 2000) + n_total
    mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (n_total, 4, 6, 2)
    assert np.array_equal(got[:, 0, 0, 0], np.arange(n_total, dtype=np.float32))


def _grad_worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
    from src.dist import allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)  # stand-in for the gradient arena of this rank
    assert allreduce_gradients(flat) == world
    if rank == 0:
        np.save(out_path, flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    out = str(tmp_path / "g.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), np.arange(1000, dtype=np.float32) * 3)


def test_allreduce_without_process_group_is_identity():
    sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
    from src.dist import allreduce_gradients
    flat = torch.ones(8)
    assert allreduce_gradients(flat) == 1 and torch.equal(flat, torch.ones(8))
