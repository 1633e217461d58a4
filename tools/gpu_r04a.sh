#!/usr/bin/env bash
# round 4, first GPU call: variants prepared at the end of round 3 WITHOUT GPU time left to run them.
#   BF_KABSCH_LANES=1   greedy Kabsch filter with the moment sums of every fit spread over lanes (bit-identical by construction; the chain test
#                       passed on the GPU in the last seconds of round 3, the kernel has not been TIMED)
# The whole GPU suite first (the parametrised chain test covers the new path), then the driver's bench with and without the variant.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04a; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
#   BF_APX_DEFER=1      fast voxel update with a pair's voxel loads issued only behind a valid sample (k_update_apx_defer; bit-identical by construction; NEVER RUN)
(timeout 200 python -m pytest tests/test_match_gpu.py -q 2>&1 | tail -6 | tee "$OUT/pytest_match.txt")
(BF_TEST_UNVERIFIED=1 timeout 200 python -m pytest tests/test_tsdf_fast_gpu.py -q -k "lds_staged and defer" 2>&1 | tail -6 | tee "$OUT/pytest_defer.txt")
for L in 1 0; do
  BF_APX_DEFER=$L timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_defer$L.json" 2> "$OUT/bench_defer$L.err" || tail -3 "$OUT/bench_defer$L.err"
  python -c "
import json; j=json.load(open('$OUT/bench_defer$L.json')); r=j['roofline']; print('defer=$L bench fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"
done
for L in 1 0; do
  BF_KABSCH_LANES=$L timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_kabsch$L.json" 2> "$OUT/bench_kabsch$L.err" || tail -3 "$OUT/bench_kabsch$L.err"
  python -c "
import json; j=json.load(open('$OUT/bench_kabsch$L.json')); r=j['roofline']; print('kabsch_lanes=$L bench fps %.1f ms/step %.3f wait_match %.3f launch_us %.1f' % (j['value'], j['ms_per_step'], j['config']['host_thread_ms_per_frame']['wait_match_result'], r['avg_launch_us']))"
done
rm -rf /tmp/r_tr; (cd /tmp && BF_KABSCH_LANES=1 timeout 200 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > /dev/null 2>&1)
python tools/rocpd_stats.py "$(ls -S /tmp/r_tr/*/*_results.db /tmp/r_tr/*_results.db 2>/dev/null | head -1)" "$OUT/kernel_stats_kabsch1.md" --exclude "Cijk_,at::native" | grep -E "kabsch|kernel" | head -4
