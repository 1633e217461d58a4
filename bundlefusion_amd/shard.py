"""Multi-GPU sharding (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).  Two modes:

* "segments" (weak scaling, bench.py default): the stream is cut into contiguous segments, one per rank; every rank runs the
  full loop on its segment into its own volume.  No data-path collective; barrier + MAX-over-ranks timing only.
* "volume-shard" (strong scaling of the volumetric half): every rank is fed the whole stream, runs the
  bit-deterministic bundling redundantly (replicated solve, 8e-3) and integrates only its hash-bucket shard of ONE volume
  (bf_scene_set_shard).  Poses need no exchange because they are bit-identical on every rank; `same_over_ranks` verifies
  exactly that with one MIN/MAX all-reduce of the trajectory after the run.
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


def same_over_ranks(tensor):
    """True iff `tensor` (same shape on every rank) holds identical bits on all ranks: MIN and MAX all-reduce of its integer
    view coincide.  Used to check that the replicated bundling of the volume-shard mode produced one trajectory."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    bits = tensor.contiguous().view(torch.int32).to(torch.int64)
    lo, hi = bits.clone(), bits.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))
