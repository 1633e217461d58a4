#!/usr/bin/env bash
# round 3, GPU call N: 8-byte {depth, colour} texel gathers in the fast voxel update (A/B) + its parity tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03n; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsdf_fast_gpu.py -q 2>&1 | tail -2
for tx in 0 1; do
  BF_APX_TEXEL=$tx BF_TSDF_ARITH=fast timeout 200 python tools/tsdf_sweep.py 2>/dev/null | python -c "import json,sys;j=json.load(sys.stdin);print('texel=$tx sweep update_us %.1f re_us %.1f'%(j['update_kernel_us_per_launch'],j['reintegrate_us_per_frame']))"
  BF_APX_TEXEL=$tx timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_t$tx.json" 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/bench_t$tx.json')); r=j['roofline']; print('texel=$tx bench fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"
done
BF_APX_TEXEL=1 BF_TSDF_ARITH=fast timeout 200 python tools/tsdf_sweep.py --width 1280 --height 960 --voxel 0.002 --frames 12 --stride 6 --buckets 4000000 --blocks 1500000 2>/dev/null | python -c "import json,sys;j=json.load(sys.stdin);print('1280x960 texel=1 update_us %.1f re_us %.1f'%(j['update_kernel_us_per_launch'],j['reintegrate_us_per_frame']))"
