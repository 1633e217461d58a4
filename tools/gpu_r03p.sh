#!/usr/bin/env bash
# round 3, GPU call P: k_update_apx_lds (footprint staged through LDS) - bit identity with the gather form, then the bench window with and without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03p; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
(timeout 100 python -m pytest tests/test_tsdf_fast_gpu.py -q -k "lds_staged and ${LDS_TEST:-3}" 2>&1 | tail -8 | tee "$OUT/pytest_lds.txt")
for L in ${LDS_LIST:-3 0}; do
  if [ $L = F ]; then export BF_APX_FULL_STORES=1 BF_APX_LDS=0; else export BF_APX_FULL_STORES=0 BF_APX_LDS=$L; fi
  timeout 90 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_lds$L.json" 2> "$OUT/bench_lds$L.err" || tail -3 "$OUT/bench_lds$L.err"
  python -c "
import json; j=json.load(open('$OUT/bench_lds$L.json')); r=j['roofline']; print('lds=$L bench fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"
done
