#!/usr/bin/env python3
"""The memory system's rate for the voxel update's ACCESS PATTERN, without its arithmetic and gathers.

bf_probe_block_copy (tools/probe/probe.hip -> tools/probe/libbf_probe.so, built by bundlefusion_amd.build.build_probe; not part of the product library): one wave per 6144-byte SDF block, blocks scattered over the heap in list order, every block read with
global_load_dwordx4 (fully coalesced 1 KB rows) and 9/12 of it written back (the update reads 6.1 KB and writes ~4.6 KB per block), launch geometry
of the update.  Two list orders: ascending block index (what a fresh heap gives: consecutive allocations) and a random permutation (a long-running
volume after garbage collections).  Heap 3.6 GB like the bench configuration (600 000 blocks), 34 000 blocks per launch.

Prints one JSON object; profiles/r04_hbm_block_probe.json holds the run on the MI355X box."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bundlefusion_amd import build as _build

lib = C.CDLL(_build.build_probe())


def check(rc):
    if rc != 0:
        raise RuntimeError("probe call failed: %d" % rc)


def main():
    heap_blocks, n, grid, reps = 600000, 34000, 8192, 50
    heap = torch.zeros(heap_blocks * 6144, dtype=torch.uint8, device="cuda")
    rng = np.random.RandomState(7)
    out = {"heap_GB": heap_blocks * 6144 / 1e9, "blocks_per_launch": n, "grid": grid, "hbm_peak_TBps": 8.0}
    for order in ("ascending", "random"):
        for name, n_blocks in (("34k", n), ("210k", 210000)):
            idx = np.sort(rng.choice(heap_blocks, n_blocks, replace=False)).astype(np.uint32)
            if order == "random":
                rng.shuffle(idx)
            lst = torch.from_numpy(idx.astype(np.int64)).to(torch.int32).cuda() if False else torch.from_numpy(idx.view(np.int32)).cuda()
            for wr in (0, 9, 12):
                us = C.c_float()
                check(lib.bf_probe_block_copy(C.c_void_p(heap.data_ptr()), C.c_void_p(lst.data_ptr()), n_blocks, wr, grid, reps, None, C.byref(us)))
                moved = n_blocks * 6144 * (1 + wr / 12.0)
                out["%s_%s_write%d_12" % (order, name, wr)] = {"us": round(us.value, 2), "TBps_read_plus_written": round(moved / (us.value * 1e-6) / 1e12, 3)}
    torch.cuda.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
