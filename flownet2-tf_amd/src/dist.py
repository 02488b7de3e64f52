"""Multi-GPU inference: image pairs are independent, so a batch shards across ranks (one
process per GPU) with NO collective on the data path; the only communication is the optional
gather of finished flow fields on rank 0.  Multi-GPU training: one all-reduce of the flat
gradient arena per step (allreduce_gradients).  The reference is single-GPU (SURVEY.md
section 2: no distributed code), so this module has no counterpart there.

torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def exchange_enabled():
    """True when gradients are exchanged: a process group with more than one rank -- or, FN2_DIST_SINGLE=1, with a
    single rank (a one-GPU rehearsal that drives the identical RCCL call sequence: communicator, bucketed
    all_reduce(async_op=True) on RCCL's stream between the captured graph segments, waits)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("FN2_DIST_SINGLE") == "1")


def shard_range(n_items, rank, world):
    """Contiguous, balanced split: the first n_items % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_flows(local_flows, n_total, dst=0):
    """Collect per-rank flow shards [n_local, H, W, 2] on rank `dst` in global pair order.
    Ragged shards are padded to the largest shard for the gather and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_flows
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    n_max = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((n_max,) + tuple(local_flows.shape[1:]), dtype=local_flows.dtype, device=local_flows.device)
    pad[:local_flows.shape[0]] = local_flows
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def allreduce_gradients(flat):
    """SUM the flat fp32 gradient arena over all ranks, in place (one collective per step: the arena is a
    single contiguous buffer, ~155 MB for FlowNetS, so the ring runs at xGMI link bandwidth rather than
    launch latency).  The 1/world mean is folded into the Adam kernel's grad_scale.  Returns world size."""
    if not exchange_enabled():
        return 1
    if flat.is_cuda and dist.get_backend() == "gloo":  # CPU-rendezvous tests of the GPU trainer
        host = flat.cpu()
        dist.all_reduce(host)
        flat.copy_(host)
    else:
        dist.all_reduce(flat)
    return dist.get_world_size()


def allreduce_bucket_async(bucket):
    """Start the SUM of one contiguous slice of the gradient arena; returns a handle with .wait() (None when
    there is nothing to wait for).  Called from inside the backward pass as soon as the slice is complete, so
    the ring runs on RCCL's stream under the remaining backward kernels (c10d orders it after the producing
    kernels of the current stream)."""
    if not exchange_enabled():
        return None
    if bucket.is_cuda and dist.get_backend() == "gloo":  # CPU-rendezvous tests: synchronous staging
        host = bucket.cpu()
        dist.all_reduce(host)
        bucket.copy_(host)
        return None
    return dist.all_reduce(bucket, async_op=True)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1
