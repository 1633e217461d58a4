#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04m; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --no-sweep $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f wait_match %.3f wait_ingest %.3f solves %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], h['wait_match_result'], h['wait_ingest'], h['solves']))"
}
timeout 900 python -m pytest tests/test_tsdf_gpu.py tests/test_tsdf_fast_gpu.py tests/test_pipeline_baseline_gpu.py tests/test_raycast_gpu.py tests/test_mesh_gpu.py -x -q -m gpu > "$OUT/pytest_part.txt" 2>&1; tail -5 "$OUT/pytest_part.txt" | cut -c1-300
BARGS="" run new BF_PIPELINE_DEPTH=2
BARGS="" run ordered BF_PIPELINE_DEPTH=2 BF_SCENE_ORDERED_LISTS=1
BARGS="" run new_b BF_PIPELINE_DEPTH=2
BARGS="" run new_d3 BF_PIPELINE_DEPTH=3
