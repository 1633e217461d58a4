#!/usr/bin/env bash
# round 3, GPU call A: fast-contract tests, voxel kernels regression, both contracts timed (driver-style bench x3 each, sweep), clocks logged
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03a; mkdir -p "$OUT"; cd "$ROOT"
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Average Graphics Package Power|Current Socket' | tr -s ' ' | tr '\n' ';')"; sleep 0.5; done ) > "$OUT/clocks.txt" 2>&1 &
CLK=$!
timeout 600 python -m pytest tests/test_tsdf_fast_gpu.py tests/test_tsdf_gpu.py -x -q -s 2>&1 | tail -25 > "$OUT/pytest_tsdf.txt"; tail -8 "$OUT/pytest_tsdf.txt"
for a in exact fast; do
  BF_TSDF_ARITH=$a timeout 200 python tools/tsdf_sweep.py 2>/dev/null > "$OUT/sweep_$a.json"; cut -c1-420 "$OUT/sweep_$a.json"
done
for i in 1 2 3; do for a in exact fast; do
  echo "== bench $a $i $(date +%s.%N)" >> "$OUT/clocks.txt"
  BF_TSDF_ARITH=$a timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_${a}_$i.json" 2> "$OUT/bench_${a}_$i.err"
  python - "$OUT/bench_${a}_$i.json" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]; print(sys.argv[1].split('/')[-1], "fps %.1f ms/step %.3f launch_us %.1f frac %.3f blocks/launch %.0f share %.2f" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["blocks_visited_per_launch"], r["share_of_step_time"]))
except Exception as e: print("bench failed", e)
PY
done; done
kill $CLK
