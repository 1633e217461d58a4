#!/usr/bin/env bash
# round 3, GPU call C: new parity tests, fast kernel pipelined vs not (+ grids), timelines (fast; exact x4 for the two modes), ATE of the product
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03c; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
line() { python - "$1" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]; print(sys.argv[1].split('/')[-1], "fps %.1f ms/step %.3f launch_us %.1f frac %.3f blocks/launch %.0f share %.2f" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["blocks_visited_per_launch"], r["share_of_step_time"]))
except Exception as e: print("bench failed", sys.argv[1], e)
PY
}
timeout 900 python -m pytest tests/test_tsdf_fast_gpu.py tests/test_tsdf_gpu.py -q -s 2>&1 | grep -E "fast contract|fast vs exact|passed|failed|Error|assert" | cut -c1-700 > "$OUT/pytest_tsdf.txt"; tail -8 "$OUT/pytest_tsdf.txt"
export BF_TSDF_ARITH=fast
for pipe in 0 1; do for grid in 8192 4096 2048; do
  BF_APX_PIPE=$pipe BF_GRID_UPDATE_COL=$grid timeout 200 python tools/tsdf_sweep.py 2>/dev/null > "$OUT/sweep_p${pipe}_g$grid.json"
  python -c "import json;j=json.load(open('$OUT/sweep_p${pipe}_g$grid.json'));print('sweep pipe=$pipe grid=$grid update_us %.1f re_us %.1f'%(j['update_kernel_us_per_launch'],j['reintegrate_us_per_frame']))"
  BF_APX_PIPE=$pipe BF_GRID_UPDATE_COL=$grid timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_p${pipe}_g$grid.json" 2>/dev/null; line "$OUT/bench_p${pipe}_g$grid.json"
done; done
# timeline of the fast contract (default variant)
rm -rf /tmp/r_tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/fast_traced.json" 2>/dev/null)
python tools/rocpd_stats.py "$(db /tmp/r_tr)" "$OUT/fast_kernel_stats.md" --exclude "Cijk_,at::native" | head -16
python tools/rocpd_timeline.py "$(db /tmp/r_tr)" 0.5 "Cijk_,at::native" > "$OUT/fast_timeline.txt" 2>&1; line "$OUT/fast_traced.json"
export BF_TSDF_ARITH=exact
for r in 1 2 3 4; do
  rm -rf /tmp/r_tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/exact_traced_$r.json" 2>/dev/null)
  python tools/rocpd_stats.py "$(db /tmp/r_tr)" "$OUT/exact_kernel_stats_$r.md" --exclude "Cijk_,at::native" > /dev/null
  python tools/rocpd_timeline.py "$(db /tmp/r_tr)" 0.5 "Cijk_,at::native" > "$OUT/exact_timeline_$r.txt" 2>&1; line "$OUT/exact_traced_$r.json"
done
unset BF_TSDF_ARITH
timeout 900 python -m pytest tests/test_pipeline_baseline_gpu.py -q -s -k "loop_closure or 1280x960" 2>&1 | grep -E "loop closure stream|1280x960|passed|failed|Error|assert" | cut -c1-600 > "$OUT/pytest_new.txt"; tail -6 "$OUT/pytest_new.txt"
timeout 300 python tools/ref_ate_table.py --side product --voxel 0.05 --buckets 50000 --blocks 40000 --out "$OUT/product.npz" 2>&1 | tail -1
timeout 300 python tools/ref_ate_table.py --side product --voxel 0.05 --buckets 50000 --blocks 40000 --exit-frames 2 --tail 4 --out "$OUT/product_dense.npz" 2>&1 | tail -1
