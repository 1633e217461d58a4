#!/usr/bin/env bash
# round 3, GPU call F: solver after the scan / max-residual rewrite (tests, scaling table), loop-closure test, 5000-frame stream, 1280x960 @2 mm sweep per contract
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03f; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
t() { name=$1; shift; timeout 420 python -m pytest "$@" -q -s -x 2>&1 | grep -E "loop closure stream|passed|failed|Error|assert" | cut -c1-900 > "$OUT/pytest_$name.txt"; echo "== $name"; tail -5 "$OUT/pytest_$name.txt"; }
t solver tests/test_solver_gpu.py
t loop tests/test_pipeline_baseline_gpu.py -k loop_closure
t pipeline tests/test_pipeline_gpu.py
timeout 400 python tools/solver_scaling.py 2> "$OUT/solver_scaling.err" | tee "$OUT/solver_scaling.md"; tail -3 "$OUT/solver_scaling.err"
export BF_TSDF_ARITH=fast
(timeout 600 python tools/run_sequence.py --frames 5000 --bob 0.3 --voxel 0.004 --buckets 4000000 --blocks 3000000 --tail 35 2>&1 | grep -E "frames|integrated|optimized|counters|allocated|rror|contract" | tee "$OUT/stream5000.txt")
for a in fast exact; do
  BF_TSDF_ARITH=$a timeout 300 python tools/tsdf_sweep.py --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 2>/dev/null > "$OUT/sweep_1280_$a.json"; cut -c1-520 "$OUT/sweep_1280_$a.json"
done
rm -rf /tmp/r_sw; (cd /tmp && BF_TSDF_ARITH=fast timeout 300 rocprofv3 --kernel-trace -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 > /dev/null 2>&1)
python tools/rocpd_stats.py "$(db /tmp/r_sw)" "$OUT/sweep_1280_kernel_stats.md" | head -12
rm -rf /tmp/r_sw; (cd /tmp && BF_TSDF_ARITH=fast timeout 300 rocprofv3 --kernel-trace -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" > /dev/null 2>&1)
python tools/rocpd_stats.py "$(db /tmp/r_sw)" "$OUT/sweep_640_kernel_stats.md" | head -12
