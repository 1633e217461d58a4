#!/usr/bin/env bash
# round 3, GPU call I: fast voxel update variants (80 SGPRs = 8 workgroups per CU; list entries through the scalar cache), frame-loop test of the fast contract, shared rendering path of bench.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03i; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
line() { python -c "
import json,sys
try:
    j=json.load(open('$1')); r=j['roofline']; print('$2', 'fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f' % (j['value'], j['ms_per_step'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))
except Exception as e: print('bench failed $2', e)
"; }
timeout 400 python -m pytest tests/test_tsdf_fast_gpu.py -q -s 2>&1 | grep -E "frame loop, fast|passed|failed|Error|assert" | cut -c1-400 | tee "$OUT/pytest_fast.txt"
export BF_TSDF_ARITH=fast
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  BF_APX_SGPR80=$1 BF_APX_SLOAD=$2 timeout 200 python tools/tsdf_sweep.py 2>/dev/null > "$OUT/sweep_s$1_l$2.json"
  python -c "import json;j=json.load(open('$OUT/sweep_s$1_l$2.json'));print('sweep sgpr80=$1 sload=$2 update_us %.1f re_us %.1f'%(j['update_kernel_us_per_launch'],j['reintegrate_us_per_frame']))"
  BF_APX_SGPR80=$1 BF_APX_SLOAD=$2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_s$1_l$2.json" 2>/dev/null; line "$OUT/bench_s$1_l$2.json" "bench sgpr80=$1 sload=$2"
done
BF_APX_SGPR80=1 BF_APX_SLOAD=1 timeout 300 python -m pytest tests/test_tsdf_fast_gpu.py -q 2>&1 | tail -2
BF_BENCH_SHARED_RENDER=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --one-contract > "$OUT/bench_shared_render.json" 2> "$OUT/bench_shared_render.err"; line "$OUT/bench_shared_render.json" "bench shared-render"; tail -2 "$OUT/bench_shared_render.err"
