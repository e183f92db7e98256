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


def bind_rank_threads(local_rank, local_world, reserve=0):
    """Give every rank of a node its own contiguous slice of the host cores this process may run on (the launcher's
    affinity mask, usually all of them) and size torch's intra-op pool to it: eight ranks that each spin up a
    256-thread pool on the same cores fight over them (image decoding, the CPU side of the launch queue).  Contiguous
    core ids keep a rank on one socket / NUMA node on the usual enumeration.  -> the list of core ids (or None where
    the platform has no sched_setaffinity).  OCC_BIND_THREADS=0 turns it off."""
    import os as _os
    if _os.environ.get("OCC_BIND_THREADS", "1") == "0" or not hasattr(_os, "sched_setaffinity") or local_world <= 1:
        return None
    avail = sorted(_os.sched_getaffinity(0))
    per = max(1, (len(avail) - reserve) // local_world)
    mine = avail[local_rank * per:(local_rank + 1) * per] or avail[-1:]
    try:
        _os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, len(mine)))
    return mine


def ddp_comm_stats(ddp_model):
    """What DistributedDataParallel's own logger measured for the last sampled iterations (it brackets the backward
    pass and every bucket's all-reduce with events): per-step backward compute time, gradient all-reduce time and how
    much of the latter ran under the former.  Enable before the timed steps with
    `ddp_model._set_ddp_runtime_logging_sample_rate(1)`.  -> dict of milliseconds (None where torch does not report it)."""
    try:
        d = ddp_model._get_ddp_logging_data()
    except Exception:
        return None
    ns = lambda k: (d.get(k) / 1e6) if isinstance(d.get(k), (int, float)) and d.get(k) >= 0 else None
    return {
        "backward_compute_ms": ns("avg_backward_compute_time"),
        "allreduce_ms": ns("avg_backward_comm_time"),
        "allreduce_overlapped_ms": ns("avg_backward_compute_comm_overlap_time"),
        "forward_compute_ms": ns("avg_forward_compute_time"),
        "bucket_cap_bytes": d.get("bucket_cap_bytes"), "num_buckets": d.get("num_buckets_reduced", d.get("num_buckets")),
        "gradient_bytes": d.get("total_parameter_size_bytes") or d.get("param_size_bytes"),
        "backend": d.get("backend_name"),
    }
