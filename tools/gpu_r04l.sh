#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04l; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f wait_match %.3f wait_ingest %.3f solves %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], h['wait_match_result'], h['wait_ingest'], h['solves']))"
}
BARGS="" run d2_tex BF_PIPELINE_DEPTH=2
BARGS="" run d2_notex BF_PIPELINE_DEPTH=2 BF_PIPELINE_TEXEL_BUDGET_GB=0
BARGS="" run d3_tex BF_PIPELINE_DEPTH=3
BARGS="" run d2_tex_b BF_PIPELINE_DEPTH=2
(cd "$ROOT/gpurun_ab/wt_3439327" && BF_PIPELINE_DEPTH=2 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_3439327.json" 2>/dev/null); python -c "
import json; j=json.load(open('$OUT/bench_3439327.json')); print('3439327 fps %.1f' % j['value'])"
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_tsdf_gpu.py tests/test_golden_ref_gpu.py tests/test_two_rank_gpu.py tests/test_tsdf_fast_gpu.py -x -q -m gpu 2>&1 | tail -4
