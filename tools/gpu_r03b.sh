#!/usr/bin/env bash
# round 3, GPU call B: fast-contract tests (all), counters of the fast voxel update (sweep), exact-mode bimodality: 5 traced bench processes + icache counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03b; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
timeout 900 python -m pytest tests/test_tsdf_fast_gpu.py tests/test_tsdf_gpu.py -q -s 2>&1 | tail -25 > "$OUT/pytest_tsdf.txt"; tail -12 "$OUT/pytest_tsdf.txt"
export BF_TSDF_ARITH=fast
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
         "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
         "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/r_sw
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" > /dev/null 2>&1)
  python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_sw)" update | grep '^|' > "$OUT/fast_sweep_pmc_pass$i.txt"; tail -3 "$OUT/fast_sweep_pmc_pass$i.txt" | cut -c1-300
done
export BF_TSDF_ARITH=exact
for r in 1 2 3 4 5; do
  rm -rf /tmp/r_tr
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/exact_bench_$r.json" 2>/dev/null)
  python "$ROOT/tools/rocpd_stats.py" "$(db /tmp/r_tr)" "$OUT/exact_kernel_stats_$r.md" --exclude "Cijk_,at::native" | head -12
  python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_tr)" update | grep '^|' > "$OUT/exact_pmc_$r.txt"; tail -2 "$OUT/exact_pmc_$r.txt" | cut -c1-300
  python - "$OUT/exact_bench_$r.json" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]; print(sys.argv[1].split('/')[-1], "fps %.1f launch_us %.1f" % (j["value"], r["avg_launch_us"]))
except Exception as e: print("bench failed", e)
PY
done
