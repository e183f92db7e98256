"""Data-parallel plumbing for the one way this path shards: by sample.

Samples (nuScenes keyframes) are independent; the model is replicated; rank r takes the samples
{i : i mod world == r} (the reference's DistributedSampler / DistributedGroupSampler role,
projects/mmdet3d_plugin/datasets/samplers/).  Inference needs no data-path collective; the only
collectives are a barrier and the MAX-reduce of the elapsed time for throughput reporting, and —
for training — DDP's gradient all-reduce (RCCL over xGMI; backend string 'nccl').
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_samples, rank, world, drop_last=False):
    """Sample indices owned by `rank`: round-robin i mod world == rank.  Without drop_last the tail
    is wrapped so every rank gets the same count (what a distributed sampler's padding does)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if drop_last:
        per = n_samples // world
        return [rank + i * world for i in range(per)]
    if n_samples == 0:
        return []
    per = (n_samples + world - 1) // world
    return [(rank + i * world) % n_samples for i in range(per)]


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over all ranks (identity when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def throughput(total_units_per_rank, elapsed_local, device=None):
    """Whole-job units/s: all ranks' units / max-over-ranks elapsed time."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return world * total_units_per_rank / max_over_ranks(elapsed_local, device)
