#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04h; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f vol %s host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], j['config']['volume_thread']['us_of_api_calls_per_operator'], h))"
}
BARGS="" run s0_nb4_d2 BF_SCENE_SPLIT_PREP=0 BF_SCENE_LIST_BUFFERS=4 BF_PIPELINE_DEPTH=2
BARGS="" run s0_nb8_d2 BF_SCENE_SPLIT_PREP=0 BF_SCENE_LIST_BUFFERS=8 BF_PIPELINE_DEPTH=2
BARGS="" run s1_nb8_d2 BF_SCENE_SPLIT_PREP=1 BF_SCENE_LIST_BUFFERS=8 BF_PIPELINE_DEPTH=2
BARGS="" run s0_nb8_d3 BF_SCENE_SPLIT_PREP=0 BF_SCENE_LIST_BUFFERS=8 BF_PIPELINE_DEPTH=3
BARGS="" run s1_nb8_d3 BF_SCENE_SPLIT_PREP=1 BF_SCENE_LIST_BUFFERS=8 BF_PIPELINE_DEPTH=3
BARGS="" run s0_nb4_d2_q4 BF_SCENE_SPLIT_PREP=0 BF_SCENE_LIST_BUFFERS=4 BF_PIPELINE_DEPTH=2 GPU_MAX_HW_QUEUES=4
