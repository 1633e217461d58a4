#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04j; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f wait_match %.3f wait_ingest %.3f solves %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], h['wait_match_result'], h['wait_ingest'], h['solves']))"
}
export BF_PIPELINE_DEPTH=2 BF_SCENE_SPLIT_PREP=0
run q4_eager GPU_MAX_HW_QUEUES=4 BF_PIPELINE_EAGER_STREAMS=1
run q8_eager GPU_MAX_HW_QUEUES=8 BF_PIPELINE_EAGER_STREAMS=1
run q4_lazy GPU_MAX_HW_QUEUES=4
run q6_lazy GPU_MAX_HW_QUEUES=6
run q2_lazy GPU_MAX_HW_QUEUES=2
run q4_eager_split GPU_MAX_HW_QUEUES=4 BF_PIPELINE_EAGER_STREAMS=1 BF_SCENE_SPLIT_PREP=1
