#!/usr/bin/env bash
# round 3, GPU call K: compute units reserved for the bundling chain (the volume stream masked off R CUs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03k; mkdir -p "$OUT"; cd "$ROOT"
line() { python -c "
import json,sys
try:
    j=json.load(open('$1')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']; print('$2', 'fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f wait %.3f solves %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], h['wait_match_result'], h['solves']))
except Exception as e: print('bench failed $2', e)
"; }
for r in 0 16 32 64 96; do
  BF_VOLUME_CU_RESERVE=$r timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_r$r.json" 2> "$OUT/bench_r$r.err"; line "$OUT/bench_r$r.json" "reserve=$r"; tail -1 "$OUT/bench_r$r.err" | cut -c1-200
done
