"""Multi-GPU sharding of the frame stream (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).

The per-frame loop has no cross-frame-chunk data dependency except through the poses, so the stream is cut into
contiguous segments, one per rank; every rank runs the full loop on its segment into its own volume.  The only
communication is the timing reduction of bench.py (barrier + MAX over ranks) — there is no data-path collective.
"""


def segment(rank, world, frames_per_rank):
    """Half-open frame range [first, last) of `rank`: contiguous, disjoint, equal length (weak scaling)."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad shard request rank=%r world=%r" % (rank, world))
    first = rank * frames_per_rank
    return first, first + frames_per_rank


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float over the default process group (identity when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank, elapsed_local, device=None):
    """Whole-job throughput: units all ranks processed / max-over-ranks elapsed time."""
    import torch.distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return world * units_per_rank / max_over_ranks(elapsed_local, device)
