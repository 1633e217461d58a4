#!/usr/bin/env bash
# round 4, second GPU call: the whole GPU suite on the restructured frame loop (two frames in flight, lagged solve, fast default), then the bench in
# the serial order and with the solves lagged
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04b; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
(timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -40 | tee "$OUT/pytest_gpu.txt")
for L in 0 10; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --solve-lag $L > "$OUT/bench_lag$L.json" 2> "$OUT/bench_lag$L.err" || tail -5 "$OUT/bench_lag$L.err"
  python -c "
import json; j=json.load(open('$OUT/bench_lag$L.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('lag=$L fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s ate %.4f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], h, j['config']['ate_rmse_vs_ground_truth_m']))"
done
