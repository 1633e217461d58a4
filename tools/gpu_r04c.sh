#!/usr/bin/env bash
# round 4: GPU suite (captured output of the passed tests kept: the fast-contract reports), then the bench: serial order with / without the side-by-side
# pair stages, and the lagged solve
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04c; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 -rP > "$OUT/pytest_gpu_full.txt" 2>&1; tail -25 "$OUT/pytest_gpu_full.txt"
grep -E "vs ORACLE|fast contract|noisy stream|frame loop, fast" "$OUT/pytest_gpu_full.txt" | cut -c1-900 > "$OUT/fast_contract_reports.txt"
run() {   # name, env..., args
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s ate %.4f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], h, j['config']['ate_rmse_vs_ground_truth_m']))"
}
BARGS="--solve-lag 0" run serial BF_X=0
BARGS="--solve-lag 0" run serial_nopair BF_PIPELINE_PAIR_STREAMS=0
BARGS="--solve-lag 10" run lag10 BF_X=0
BARGS="--solve-lag 5" run lag5 BF_X=0
BARGS="--solve-lag 0 --steps 200" run serial200 BF_X=0
