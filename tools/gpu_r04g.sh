#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04g; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f vol %s host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], j['config']['volume_thread'], h))"
}
BARGS="" run d3 BF_PIPELINE_DEPTH=3
BARGS="" run d3_b BF_PIPELINE_DEPTH=3
timeout 120 python tools/ref_ate_table.py --side product --out "$OUT/product.npz" 2>&1 | tail -2
