#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on the voxel update's OWN access widths: MI355X_MICROARCH.md's x2 correction of
FETCH_SIZE is measured on 16-byte-per-lane loads; the update issues dword loads 12 bytes apart (three per 768-byte slice) and dword stores.

    run   (under rocprofv3, one counter per pass):   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o run -- python tools/pmc_calibrate.py run
                                                     rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -o run -- python tools/pmc_calibrate.py run
    read  python tools/pmc_calibrate.py read <fetch_db> <write_db> [out.json]

`run` launches, on a 3.7 GB heap (beyond the 256 MB memory-side cache), REPS launches each of
   k_probe_blocks  (global_load_dwordx4 rows, 9/12 written back)     over 210 000 blocks  -> known bytes read 6144 / block, written 4608 / block
   k_probe_slices  (the update's dword pattern, 6 of 8 slices written) over 210 000 blocks -> known bytes read 6144 / block, written 4608 / block
`read` divides the known bytes by the counters (KiB) and prints the factor each counter has to be multiplied with on that pattern."""
import ctypes as C
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HEAP_BLOCKS, N, GRID, REPS = 600000, 210000, 8192, 20


def run():
    import numpy as np
    import torch
    from bundlefusion_amd import build
    lib = C.CDLL(build.build_probe())
    heap = torch.zeros(HEAP_BLOCKS * 6144, dtype=torch.uint8, device="cuda")
    rng = np.random.RandomState(7)
    idx = rng.choice(HEAP_BLOCKS, N, replace=False).astype(np.uint32)          # random order: a long-running volume's list
    lst = torch.from_numpy(idx.view(np.int32)).cuda()
    us = C.c_float()
    assert lib.bf_probe_block_copy(C.c_void_p(heap.data_ptr()), C.c_void_p(lst.data_ptr()), N, 9, GRID, REPS, None, C.byref(us)) == 0
    assert lib.bf_probe_slices(C.c_void_p(heap.data_ptr()), C.c_void_p(lst.data_ptr()), N, 6, GRID, REPS, None) == 0
    torch.cuda.synchronize()
    print(json.dumps({"blocks": N, "reps": REPS, "dwordx4_us": us.value}))


def totals(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        for k in ("k_probe_blocks", "k_probe_slices"):
            if k in name:
                out[k] = (n, tot)
    return out


def read(fdb, wdb, outp=None):
    F, W = totals(fdb, "FETCH_SIZE"), totals(wdb, "WRITE_SIZE")
    res = {"blocks_per_launch": N, "known_read_bytes_per_launch": N * 6144, "known_written_bytes_per_launch": N * 4608,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (own passes) around `python tools/pmc_calibrate.py run`; heap 3.7 GB, 210 000 blocks per launch in random order"}
    for k, what in (("k_probe_blocks", "global_load_dwordx4 / global_store_dwordx4 rows (16 bytes per lane)"), ("k_probe_slices", "the voxel update's pattern: dword loads / stores 12 bytes apart, three per 768-byte slice")):
        nf, f = F.get(k, (0, 0.0)); nw, w = W.get(k, (0, 0.0))
        fb, wb = 1024.0 * f / max(nf, 1), 1024.0 * w / max(nw, 1)          # counter bytes per launch (KiB -> bytes)
        res[k] = {"pattern": what, "launches_fetch_pass": nf, "launches_write_pass": nw, "FETCH_SIZE_bytes_per_launch": fb, "WRITE_SIZE_bytes_per_launch": wb,
                  "fetch_factor": N * 6144 / fb if fb else None, "write_factor": N * 4608 / wb if wb else None}
    print(json.dumps(res, indent=1))
    if outp:
        json.dump(res, open(outp, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        read(*sys.argv[2:5])
