#!/usr/bin/env bash
# round 4: two preparation streams (allocation / lists) with 8 hardware queues; A/B against the default 4 queues; loop depth; trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04f; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tsdf_gpu.py tests/test_pipeline_gpu.py tests/test_golden_ref_gpu.py -x -q -m gpu --durations=5 -rP > "$OUT/pytest_part.txt" 2>&1; tail -8 "$OUT/pytest_part.txt"; grep -E "vs the REFERENCE|integrations /" "$OUT/pytest_part.txt" | cut -c1-500
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract $BARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || tail -5 "$OUT/bench_$name.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$name.json')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$name fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], h))"
}
BARGS="" run q8_d3 BF_PIPELINE_DEPTH=3
BARGS="" run q8_d2 BF_PIPELINE_DEPTH=2
BARGS="" run q16_d3 BF_PIPELINE_DEPTH=3 GPU_MAX_HW_QUEUES=16
BARGS="" run q4_d3 BF_PIPELINE_DEPTH=3 GPU_MAX_HW_QUEUES=4
BARGS="--solve-lag 10" run q8_d3_lag10 BF_PIPELINE_DEPTH=3
BARGS="--solve-lag 10" run q8_d3_lag10_pair BF_PIPELINE_DEPTH=3 BF_PIPELINE_PAIR_STREAMS=1
rm -rf /tmp/r_tr; (cd /tmp && BF_PIPELINE_DEPTH=3 timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_traced.json" 2>/dev/null)
D=$(ls -S /tmp/r_tr/*/*_results.db /tmp/r_tr/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$D" "$OUT/kernel_stats.md" --exclude "Cijk_,at::native" | head -12
python tools/rocpd_timeline.py "$D" 0.8 "Cijk_,at::native" > "$OUT/timeline.txt" 2>&1; grep -E "^queue|GPU busy|k_update_apx -> |k_alloc|k_compact" "$OUT/timeline.txt" | head -40
