#!/usr/bin/env python3
"""Turn the two PMC passes of `bench.py --pmc-out acc.json` (rocprofv3 --kernel-trace --pmc FETCH_SIZE, and --pmc WRITE_SIZE, each its
own run) into profiles/rNN_pmc_tsdf_update.json: HBM bytes per visited SDF block for the plain and for the fused voxel-update kernel.

    python tools/pmc_to_json.py <fetch_db> <write_db> <acc.json> <out.json> [out.md] [arith] [calibration.json] [sq_db]

The output file holds one entry per arithmetic contract of the voxel update ("fast" / "exact", bench.py --arith); a run adds or replaces its own.

FETCH_SIZE / WRITE_SIZE are in KiB.  FETCH_SIZE under-reports on gfx950 (MI355X_MICROARCH.md: half the bytes of a wide coalesced read stream); the factor applied
is the one MEASURED with known byte counts on the voxel update's own access pattern (tools/pmc_calibrate.py -> calibration.json: k_probe_slices, 12-byte
voxels in 768-byte slices: 2.65; 16 bytes per lane: 1.99; WRITE_SIZE: 1.00) when that file is given, the guide's 2.0 otherwise.  Both factors are recorded.
The accounting file is what the profiled run itself counted over ALL its launches."""
import hashlib
import json
import os
import sqlite3
import sys


def update_kernel_sha():
    """sha256 of the voxel-update section of csrc/tsdf.hip (from the column kernel's header to the cvt probe): bench.py refuses PMC figures collected on another
    version of these kernels (the committed per-block traffic went stale silently when the kernel changed)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bundlefusion_amd", "csrc", "tsdf.hip")).read()
    a, b = src.index("// voxel update, column form: ONE WAVE per SDF block"), src.index("__global__ void k_probe_cvt")
    batch = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bundlefusion_amd", "csrc", "tsdf_batch.h")).read()
    return hashlib.sha256((src[a:b] + batch[batch.index("// the batch's voxel update, fast contract"):]).encode()).hexdigest()


def build_flags_sha():
    """sha256 of the library's compiler flags (bundlefusion_amd/build.py HIP_FLAGS): counters of a binary built with other flags (round 5: packed FP32 on / off)
    are another kernel's counters even when the source is the same"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bundlefusion_amd.build import HIP_FLAGS
    return hashlib.sha256(" ".join(HIP_FLAGS).encode()).hexdigest()


def sq_totals(db):
    """per counter: (launches, sum) over the union-list launches of the voxel update (an SQ pass: SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU ...)"""
    c = sqlite3.connect(db)
    out = {}
    for name, counter, n, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        if "k_update_batch" in name or "k_update_apx<2" in name or "k_update_col<2>" in name:
            a = out.setdefault(counter, [0, 0.0]); a[0] += n; a[1] += tot
    return out


def totals(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        fused = "k_reupdate" in name or "k_update_col<2>" in name or "k_update_apx<2" in name or "k_update_batch" in name      # union-list launches: fused re-integration, batch
        plain = not fused and ("k_update<" in name or "k_update_col<" in name or "k_update_apx<" in name)
        key = "fused" if fused else ("plain" if plain else None)
        if key:
            a = out.setdefault(key, [0, 0.0]); a[0] += n; a[1] += tot
    return out


def main():
    fdb, wdb, accp, outp = sys.argv[1:5]
    acc = json.load(open(accp))
    F, Wr = totals(fdb, "FETCH_SIZE"), totals(wdb, "WRITE_SIZE")
    res = {"config": acc["config"], "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) around `python bench.py --no-cpu-baseline --pmc-out ...`; "
                                              "FETCH_SIZE x calibrated factor, KiB -> bytes"}
    vis = {"fused": acc["visited_blocks_fused"], "plain": acc["visited_blocks_plain"]}
    lau = {"fused": acc["fused_launches"], "plain": acc["launches"] - acc["fused_launches"]}
    arith = sys.argv[6] if len(sys.argv) > 6 else "fast"
    cal = json.load(open(sys.argv[7])) if len(sys.argv) > 7 and os.path.exists(sys.argv[7]) else None
    ff = cal["k_probe_slices"]["fetch_factor"] if cal else 2.0
    wf = cal["k_probe_slices"]["write_factor"] if cal else 1.0
    res["fetch_factor_applied"] = ff; res["write_factor_applied"] = wf
    res["calibration"] = ({"k_probe_slices": cal["k_probe_slices"], "k_probe_blocks": cal["k_probe_blocks"]} if cal else "none given: the guide's x2")
    lines = ["| kernel | launches (PMC pass / run accounting) | FETCH_SIZE x factor [MB/launch] | WRITE_SIZE [MB/launch] | HBM bytes / visited block | algorithmic bytes / block |", "|---|---|---|---|---|---|"]
    for k in ("fused", "plain"):
        n, f = F.get(k, [0, 0.0]); n2, w = Wr.get(k, [0, 0.0])
        fb, wb = ff * f * 1024.0, wf * w * 1024.0
        res[k] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n2, 1),
                  "hbm_bytes_per_launch": fb / max(n, 1) + wb / max(n2, 1), "visited_blocks": vis[k],
                  "hbm_bytes_per_visited_block": (fb + wb) / max(vis[k], 1)}
        lines.append("| voxel update, %s contract, `%s` | %d / %d | %.1f | %.1f | %.0f | %d |" % (arith, k, n, lau[k], fb / max(n, 1) / 1e6, wb / max(n2, 1) / 1e6, res[k]["hbm_bytes_per_visited_block"], 512 * 24 + 32))
    sqdb = sys.argv[8] if len(sys.argv) > 8 and os.path.exists(sys.argv[8]) else None
    if sqdb:          # the SQ pass of the same command: vector instructions issued and busy cycles of the union-list launches
        sq = sq_totals(sqdb)
        n = max(sq.get("SQ_INSTS_VALU", [0, 0.0])[0], 1)
        res["sq"] = {"launches": n, "per_launch": {k: v[1] / max(v[0], 1) for k, v in sq.items()},
                     "valu_wave_instructions_per_visited_block": sq.get("SQ_INSTS_VALU", [0, 0.0])[1] / max(vis["fused"], 1),
                     "valu_active_cycles_per_visited_block": 4.0 * sq.get("SQ_ACTIVE_INST_VALU", [0, 0.0])[1] / max(vis["fused"], 1),      # (the counter is in units of 4 cycles)
                     "note": "SQ_INSTS_VALU counts wave-instructions; the cycle counters are in units of 4 cycles (MI355X_MICROARCH.md: SQ)"}
    mem, memtot = {}, {}
    for a in sys.argv[9:11]:          # the memory-pipeline passes (TD_* / TCP_* / GRBM_GUI_ACTIVE, SQ_INSTS_VMEM_* ...) of the same command, when given
        if os.path.exists(a):
            for k, v in sq_totals(a).items():
                mem[k] = v[1] / max(v[0], 1); memtot[k] = v[1]
    if mem:
        cus = 256 - max(0, min(int(os.environ.get("BF_VOLUME_CU_RESERVE", "32")), 255))      # the volume stream's CU mask (bf_pipeline_create)
        dev_cycles = mem.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                                    # summed over the 8 XCDs
        vmem = mem.get("SQ_INSTS_VMEM_RD", 0.0) + mem.get("SQ_INSTS_VMEM_WR", 0.0)
        share = (lambda name: mem.get(name, 0.0) / (dev_cycles * cus) if dev_cycles > 0 else None)
        res["mem_pipe"] = {"per_launch": mem, "compute_units": cus, "device_cycles_per_launch": dev_cycles,
                           "td_busy_share_of_cu_cycles": share("TD_TD_BUSY_sum"), "td_stalled_on_l1_share_of_cu_cycles": share("TD_TC_STALL_sum"),
                           "l1_accesses_per_cu_cycle": share("TCP_TOTAL_CACHE_ACCESSES_sum"),
                           "l1_accesses_per_vmem_instruction": mem.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / vmem if vmem > 0 else None,
                           "td_busy_cycles_per_visited_block": memtot.get("TD_TD_BUSY_sum", 0.0) / max(vis["fused"], 1),
                           "vmem_wave_instructions_per_visited_block": (memtot.get("SQ_INSTS_VMEM_RD", 0.0) + memtot.get("SQ_INSTS_VMEM_WR", 0.0)) / max(vis["fused"], 1),
                           "note": "TD_*: the texture-data (vector-memory return) unit of every CU, cycles summed over the CUs; the shares are over the CUs the masked volume stream may use "
                                   "and the device cycles of the same pass (GRBM_GUI_ACTIVE / 8 XCDs)"}
    allres = json.load(open(outp)) if os.path.exists(outp) else {}
    res["update_kernel_sha256"] = update_kernel_sha()
    res["build_flags_sha256"] = build_flags_sha()
    allres[arith] = res
    json.dump(allres, open(outp, "w"), indent=1)
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 5:
        open(sys.argv[5], "a").write(text + "\n")


if __name__ == "__main__":
    main()
