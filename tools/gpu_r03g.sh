#!/usr/bin/env bash
# round 3, GPU call G: allocation changes (one workgroup per bin; collect / ingest), solver scaling table, 2 mm sweep again
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03g; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
t() { name=$1; shift; timeout 420 python -m pytest "$@" -q -s -x 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-700 > "$OUT/pytest_$name.txt"; echo "== $name"; tail -5 "$OUT/pytest_$name.txt"; }
t tsdf tests/test_tsdf_gpu.py tests/test_tsdf_fast_gpu.py
t baseline tests/test_pipeline_baseline_gpu.py -k "three_chunks or resample or 1280x960"
timeout 400 python tools/solver_scaling.py 2> "$OUT/solver_scaling.err" | tee "$OUT/solver_scaling.md"; tail -3 "$OUT/solver_scaling.err"
export BF_TSDF_ARITH=fast
timeout 300 python tools/tsdf_sweep.py --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 2>/dev/null > "$OUT/sweep_1280_fast.json"; cut -c1-420 "$OUT/sweep_1280_fast.json"
timeout 300 python tools/tsdf_sweep.py --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 --shard-alloc 2>/dev/null > "$OUT/sweep_1280_fast_shardalloc.json"; cut -c1-420 "$OUT/sweep_1280_fast_shardalloc.json"
rm -rf /tmp/r_sw; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 --shard-alloc > /dev/null 2>&1)
python tools/rocpd_stats.py "$(db /tmp/r_sw)" "$OUT/sweep_1280_shardalloc_kernel_stats.md" | head -12
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_fast.json" 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/bench_fast.json')); r=j['roofline']; print('bench fast fps %.1f launch_us %.1f frac %.3f traffic %s hbm_frac %s share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['traffic'], r['hbm_frac_measured'], r['share_of_step_time']))"
