#!/usr/bin/env bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03m; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsdf_fast_gpu.py -q 2>&1 | tail -2
BF_TSDF_ARITH=fast timeout 200 python tools/tsdf_sweep.py 2>/dev/null | python -c "import json,sys;j=json.load(sys.stdin);print('sweep update_us %.1f re_us %.1f'%(j['update_kernel_us_per_launch'],j['reintegrate_us_per_frame']))"
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_$i.json" 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/bench_$i.json')); r=j['roofline']; print('bench fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"; done
