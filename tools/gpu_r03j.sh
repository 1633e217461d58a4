#!/usr/bin/env bash
# round 3, GPU call J: the matching chain enqueued in the delivering call (A/B), pipeline parity tests on it, SQ counters of the feature-pipeline kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03j; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
line() { python -c "
import json,sys
try:
    j=json.load(open('$1')); r=j['roofline']; print('$2', 'fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time'], j['config']['host_thread_ms_per_frame']))
except Exception as e: print('bench failed $2', e)
"; }
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_golden_ref_gpu.py tests/test_pipeline_baseline_gpu.py tests/test_evaluator_gpu.py tests/test_two_rank_gpu.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_pipeline.txt"
for e in 0 1 0 1; do
  BF_PIPELINE_EARLY_CHAIN=$e timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_e$e.json" 2>/dev/null; line "$OUT/bench_e$e.json" "bench early_chain=$e"
done
BF_PIPELINE_EARLY_CHAIN=1 timeout 300 python bench.py --no-cpu-baseline --one-contract > "$OUT/bench200_e1.json" 2>/dev/null; line "$OUT/bench200_e1.json" "bench 200 steps early_chain=1"
BF_PIPELINE_EARLY_CHAIN=0 timeout 300 python bench.py --no-cpu-baseline --one-contract > "$OUT/bench200_e0.json" 2>/dev/null; line "$OUT/bench200_e0.json" "bench 200 steps early_chain=0"
rm -rf /tmp/r_sq; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/r_sq -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > /dev/null 2>&1)
python tools/rocpd_pmc.py "$(db /tmp/r_sq)" "" | grep -E "k_filter_kabsch|k_blur|k_keys_finalize|k_match|k_filter_surface|k_filter_dense|k_cache_geometry|k_fuse_to_global|k_descriptor|k_orientation|k_add_residuals" > "$OUT/sq_feature_kernels.txt"; wc -l "$OUT/sq_feature_kernels.txt"
