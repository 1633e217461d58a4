#!/usr/bin/env bash
# round 4: the operator preparation on three streams - volume tests, then the bench (loop depth 2 / 3, serial and lagged), trace of the best
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04e; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tsdf_gpu.py tests/test_tsdf_fast_gpu.py tests/test_pipeline_baseline_gpu.py tests/test_pipeline_gpu.py tests/test_solver_gpu.py tests/test_two_rank_gpu.py -x -q -m gpu --durations=5 -rP > "$OUT/pytest_part.txt" 2>&1; tail -12 "$OUT/pytest_part.txt"
grep -E "^N = " "$OUT/pytest_part.txt" | cut -c1-400
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], h))"
}
BARGS="" run d2 BF_PIPELINE_DEPTH=2
BARGS="" run d3 BF_PIPELINE_DEPTH=3
BARGS="--solve-lag 10" run d3_lag10 BF_PIPELINE_DEPTH=3
BARGS="--arith exact" run d3_exact BF_PIPELINE_DEPTH=3
rm -rf /tmp/r_tr; (cd /tmp && BF_PIPELINE_DEPTH=3 timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_traced.json" 2>/dev/null)
D=$(ls -S /tmp/r_tr/*/*_results.db /tmp/r_tr/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$D" "$OUT/kernel_stats.md" --exclude "Cijk_,at::native" | head -12
python tools/rocpd_timeline.py "$D" 0.8 "Cijk_,at::native" > "$OUT/timeline.txt" 2>&1; grep -E "^queue|GPU busy|k_update_apx -> |k_alloc" "$OUT/timeline.txt" | head -40
