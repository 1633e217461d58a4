"""Multi-GPU sharding (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).  Two modes:

* "segments" (weak scaling, bench.py default): the stream is cut into contiguous segments, one per rank; every rank runs the
  full loop on its segment into its own volume.  No data-path collective; barrier + MAX-over-ranks timing only.
* "volume-shard" (strong scaling of the volumetric half): every rank is fed the whole stream, runs the
  bit-deterministic bundling redundantly (replicated solve, 8e-3) and integrates only its hash-bucket shard of ONE volume
  (bf_scene_set_shard).  Poses need no exchange because they are bit-identical on every rank; `same_over_ranks` verifies
  exactly that with one MIN/MAX all-reduce of the trajectory after the run.

* "chunks" (strong scaling of ONE stream, SURVEY.md 8e-1 + 8e-2 + 8e-3; bench.py --gpus N default): local chunks are dealt
  round-robin to the ranks (`chunk_owner`); each rank runs the chunk-local half (SIFT, matching inside the chunk, local solve,
  key-frame fusion: capi.ChunkWorker) for its chunks; one all-gather per round of `world` chunks hands every package to every rank
  (`gather_packages`: RCCL over xGMI, ~0.4 MB per chunk); then every rank runs the global half on all packages in stream order and
  integrates into its hash-bucket shard of the one volume (`run_chunked`).  The global solve is replicated (it is deterministic, so
  the pose update needs no further exchange; `same_over_ranks` checks it).
"""


def chunk_owner(chunk, world, first_chunk=0):
    """Rank that runs the chunk-local half of local chunk `chunk`: round robin inside rounds of `world` consecutive chunks."""
    return (chunk - first_chunk) % world


def chunk_frames(chunk, submap):
    """Stream frames [first, last] of local chunk `chunk`: submap + 1 frames, the first shared with the previous chunk."""
    return chunk * submap, chunk * submap + submap


def timed_window(preroll, steps, world, submap=10):
    """Frame counts of a measurement over the chunk-parallel mode: (pre, total, stream_length).

    `pre` untimed frames (>= preroll, rounded up so that the timed window starts exactly at a round boundary: frame 0 + whole rounds of
    world * submap frames), `total` = pre + steps, and the stream length the runner needs.  The local half of round R + 1 runs while round R goes
    through the global half, so the stream must hold one round MORE than the one the window ends in - otherwise a timed window contains the
    replicated global half and the volume only (the local halves of its own round ran before it started)."""
    rnd = world * submap
    pre = ((max(preroll, 1) - 1 + rnd - 1) // rnd) * rnd + 1
    total = pre + steps
    last_chunk = (total - 2) // submap
    return pre, total, ((last_chunk // world + 2) * world) * submap + 1


def gather_packages(mine, world, rank, device=None):
    """All-gather of one round of chunk packages: `mine` (uint8 numpy array, zeros when this rank had no chunk in the round) ->
    list of `world` arrays indexed by owner rank.  One collective per round; identity for world == 1."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return [mine]
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.from_numpy(mine).to(dev)
    out = torch.empty((world, t.numel()), dtype=torch.uint8, device=dev)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, t)                 # one RCCL all-gather (ring over xGMI)
    else:
        dist.all_gather(list(out.unbind(0)), t)             # gloo (CPU tests)
    host = out.cpu().numpy()
    return [np.ascontiguousarray(host[r]) for r in range(world)]


class ChunkedRunner:
    """Drives one rank of the chunk-parallel mode.  `pipe` has its volume sharded with set_volume_shard(rank, world); `feed` is the whole
    stream as (depth, colour) cuda tensors.  advance(n) pushes the next n frames of the stream through the global half; whenever a frame
    of a local chunk without a package is reached, the round of `world` chunks containing it is produced: local half of this rank's
    chunk of the round (capi.ChunkWorker), then ONE all-gather.  Every rank must call advance() with the same arguments."""

    def __init__(self, pipe, worker, feed, submap, rank=0, world=1, device=None, prefetch=True, comm=None):
        self.pipe, self.worker, self.feed, self.S = pipe, worker, feed, submap
        if hasattr(pipe, "set_solve_lag"):
            pipe.set_solve_lag(0)       # the replicated global half runs in the serial order on every rank (the pipeline's default is the lagged schedule)
        self.rank, self.world, self.device = rank, world, device
        self.comm = comm                # capi.Comm: the round's all-gather through the C ABI (bf_chunk_exchange: RCCL, or the host's callback) instead of torch.distributed
        self.next_frame = 0
        self.pkgs = {}
        self.rounds = 0
        self.local_chunks = 0           # packages of this rank handed to an all-gather
        self.local_runs = 0             # chunk-local halves this rank has RUN (counted when the run finishes, on whichever thread ran it)
        # the local half of this rank's chunk of the NEXT round runs on a second host thread (own handles, own stream) while the main
        # thread pushes the current round through the global half; collectives stay on the main thread
        self.prefetch = prefetch
        self._pending = None            # (round_start, thread, package, produced)
        # the current HIP device is per host thread: the thread running ahead has to select this rank's GPU itself (on a node where every
        # process sees all GPUs it would otherwise work on device 0 with handles that live on device `rank`)
        self._device_index = None
        if device is not None and str(device).startswith("cuda"):
            import torch
            self._device_index = torch.cuda.current_device()

    def frames_needed(self, upto_frame):
        """Stream length needed to advance to `upto_frame` frames: the last round's chunks must be complete."""
        last_chunk = 0 if upto_frame <= 1 else (upto_frame - 2) // self.S
        round_end = (last_chunk // self.world + 1) * self.world
        return round_end * self.S + 1

    def _local_half(self, r0, out, produced):
        try:
            if self._device_index is not None:
                import torch
                torch.cuda.set_device(self._device_index)
            c_mine = r0 + self.rank
            a, b = chunk_frames(c_mine, self.S)
            if b < len(self.feed):
                self.worker.run(c_mine, self.feed[a:b + 1], out=out)
                produced.append(c_mine)
                self.local_runs += 1
        except BaseException as e:           # raised again on the main thread when the package is needed (wait / _ensure)
            produced.append(e)

    def _start(self, r0):
        import threading
        import numpy as np
        pkg = np.zeros(self.worker.package_bytes, np.uint8)
        produced = []
        if self.prefetch:
            t = threading.Thread(target=self._local_half, args=(r0, pkg, produced))
            t.start()
        else:
            t = None
        self._pending = (r0, t, pkg, produced)

    def _ensure(self, chunk):
        if chunk in self.pkgs:
            return
        r0 = chunk - chunk % self.world
        if self._pending is None or self._pending[0] != r0:
            if self._pending is not None and self._pending[1] is not None:
                self._pending[1].join()
            self._start(r0)
        _, t, mine, produced = self._pending
        if t is not None:
            t.join()
        else:
            self._local_half(r0, mine, produced)
        self._pending = None
        for p in produced:
            if isinstance(p, BaseException):
                raise p
        self.local_chunks += len(produced)
        got = self.comm.chunk_exchange(mine) if self.comm is not None else gather_packages(mine, self.world, self.rank, self.device)
        self.rounds += 1
        for i, p in enumerate(got):
            self.pkgs[r0 + i] = p
        if self.prefetch:
            nxt = r0 + self.world
            a, b = chunk_frames(nxt + self.rank, self.S)
            if b < len(self.feed):
                self._start(nxt)

    def wait(self):
        """Block until the local half running ahead (if any) has finished; its package is kept for the round that needs it."""
        if self._pending is not None and self._pending[1] is not None:
            self._pending[1].join()
            for p in self._pending[3]:
                if isinstance(p, BaseException):
                    raise p

    def close(self):
        self.wait()
        self._pending = None

    def advance(self, n):
        for f in range(self.next_frame, self.next_frame + n):
            c = 0 if f == 0 else (f - 1) // self.S
            self._ensure(c)
            ok = self.pipe.process_frame_chunked(self.feed[f][0], self.feed[f][1], self.pkgs[c], f - c * self.S)      # not inside an assert: python -O strips those
            if not ok:
                raise RuntimeError("frame %d of the stream was not accepted by the global half" % f)
            for old in [k for k in self.pkgs if k < c - self.world]:
                del self.pkgs[old]
        self.next_frame += n
        return n


def run_chunked(pipe, worker, feed, submap, rank=0, world=1, device=None, prefetch=True):
    """The whole stream `feed` (1 + k * submap frames) through ChunkedRunner; returns the number of frames processed."""
    r = ChunkedRunner(pipe, worker, feed, submap, rank, world, device, prefetch)
    n = r.advance(len(feed))
    r.close()
    return n


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
