#!/usr/bin/env bash
# One GPU call = the standard measurements of a round, written under gpurun_out/rNN/ (copy what is to be judged into profiles/).
#   usage (on the GPU box, from the repository root):  bash tools/gpu_round.sh <NN> [tests] [bench] [trace] [pmc] [sq] [stream] [sens]
#   e.g.  gpurun --timeout 1500 -- 'bash tools/gpu_round.sh 02 tests bench trace'
# Every step has its own timeout; PMC passes are separate rocprofv3 runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; never combine --pmc with the sys/hip/hsa trace domains).  rocprofv3 runs from /tmp with TMPDIR=/tmp.
set -u
NN=${1:-00}; shift || true
STEPS=${*:-tests bench trace}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r$NN
SWEEP_ARGS=${SWEEP_ARGS:-}
BENCH_ARGS=${BENCH_ARGS:-}
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
for step in $STEPS; do
  case $step in
    tests)      # the whole GPU suite; the captured output of passed tests (the fast-contract reports, the reference-fixture comparison) is kept
      (cd "$ROOT" && timeout 1100 python -m pytest tests -q -m gpu --durations=12 -rP > "$OUT/pytest_gpu_full.txt" 2>&1; tail -22 "$OUT/pytest_gpu_full.txt" | tee "$OUT/pytest_gpu.txt"
       grep -E "^E  " "$OUT/pytest_gpu_full.txt" | head -40
       grep -E "vs ORACLE|vs the REFERENCE|fast contract|noisy stream|frame loop, fast|^N = |integrations /|configs\[2\] at length" "$OUT/pytest_gpu_full.txt" | cut -c1-900 > "$OUT/test_reports.txt") ;;
    tests_sel)  # a selection of the GPU suite: TESTS="tests/test_tsdf_gpu.py tests/test_pipeline_gpu.py"
      (cd "$ROOT" && timeout 900 python -m pytest ${TESTS:-tests/test_tsdf_gpu.py tests/test_tsdf_fast_gpu.py tests/test_pipeline_gpu.py} -q -m gpu --durations=5 > "$OUT/pytest_sel_full.txt" 2>&1; grep -E "^E  |Error|assert" "$OUT/pytest_sel_full.txt" | head -30; tail -12 "$OUT/pytest_sel_full.txt" | tee "$OUT/pytest_sel.txt") ;;
    smoke)
      (cd "$ROOT" && timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee "$OUT/smoke.txt") ;;
    long)       # BASELINE configs[2] (2000-frame loop closure) and configs[3] (5000 frames) at full length through bench.py's long_stream block
      for N in ${LONG_FRAMES:-2000 5000}; do
        (cd "$ROOT" && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --no-sweep --long-stream $N > "$OUT/bench_long_$N.json" 2> "$OUT/bench_long_$N.err"; python -c "
import json; j=json.load(open('$OUT/bench_long_$N.json'))['long_stream']; print({k: j[k] for k in ('frames','value','last_over_first','frames_tracked','ate_integrated_m','ate_optimized_m','counters')})"; tail -2 "$OUT/bench_long_$N.err")
      done ;;
    probe)
      (cd "$ROOT" && timeout 200 python tools/hbm_block_probe.py > "$OUT/hbm_block_probe.json" 2>/dev/null; cut -c1-600 "$OUT/hbm_block_probe.json") ;;
    tests_new)
      (cd "$ROOT" && timeout 900 python -m pytest tests/test_pipeline_baseline_gpu.py -x -q --durations=8 2>&1 | tail -25 | tee "$OUT/pytest_new.txt") ;;
    bench)
      (cd "$ROOT" && timeout 400 python bench.py $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-260 "$OUT/bench.json"; tail -3 "$OUT/bench.err") ;;
    bench_driver)   # the driver's invocation
      (cd "$ROOT" && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; cut -c1-260 "$OUT/bench_driver.json"; tail -3 "$OUT/bench_driver.err") ;;
    bench_env)      # the driver's window under environment variants: ENVS="A=1;B=2 C=3" runs once per ';'-separated assignment list
      IFS=';' read -ra VARIANTS <<< "${ENVS:-}"
      for V in "${VARIANTS[@]}"; do
        TAG=$(echo "$V" | tr ' =/' '___' | tail -c 80)
        (cd "$ROOT" && env $V timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --no-sweep --long-stream 0 $BENCH_ARGS > "$OUT/bench_env_$TAG.json" 2> "$OUT/bench_env_$TAG.err"; python -c "
import json; j=json.load(open('$OUT/bench_env_$TAG.json')); print('$V', round(j['value'],1), 'fps', 'update', round(j['roofline']['avg_launch_us'],1), 'us', j['config']['host_thread_ms_per_frame'])"; tail -1 "$OUT/bench_env_$TAG.err")
      done ;;
    bench_ab)       # the driver's window with the frame's TSDF operators batched (default) and one at a time: same code, same box
      for B in on off; do
        (cd "$ROOT" && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --no-sweep --long-stream 0 --volume-batching $B $BENCH_ARGS > "$OUT/bench_batching_$B.json" 2> "$OUT/bench_batching_$B.err"; python -c "
import json; j=json.load(open('$OUT/bench_batching_$B.json')); r=j['roofline']; print('$B', round(j['value'],1), 'fps', {k: (round(r[k],3) if isinstance(r[k], float) else r[k]) for k in ('avg_launch_us','us_per_operator','frac','frac_per_operator','launches','operators','blocks_visited_per_launch')}, j['config']['host_thread_ms_per_frame'], j['config']['volume_thread'])"; tail -2 "$OUT/bench_batching_$B.err")
      done ;;
    trace)
      rm -rf /tmp/r_trace
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/r_trace -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract $BENCH_ARGS > "$OUT/bench_traced.json" 2> /dev/null)
      D=$(db /tmp/r_trace)
      rm -f "$OUT/kernel_stats.md"
      python "$ROOT/tools/rocpd_stats.py" "$D" "$OUT/kernel_stats.md" --exclude "Cijk_,at::native" | head -14
      python "$ROOT/tools/rocpd_timeline.py" "$D" 0.8 "Cijk_,at::native" > "$OUT/timeline.txt" 2>&1; tail -1 "$OUT/timeline.txt"
      python "$ROOT/tools/rocpd_timeline.py" "$D" 0.8 "Cijk_,at::native" --dump ${DUMP_MS:-4} --dump-end ${DUMP_END_MS:-6} > "$OUT/timeline_window.txt" 2>&1; wc -l "$OUT/timeline_window.txt" ;;
    pmc)   # HBM traffic of the voxel update in the bench configuration, per arithmetic contract: two passes (FETCH_SIZE, WRITE_SIZE), then bytes per visited block
      for A in ${PMC_CONTRACTS:-fast exact}; do
        for C in FETCH_SIZE WRITE_SIZE; do
          rm -rf /tmp/r_pmc_$C
          (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_pmc_$C -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract --arith $A $BENCH_ARGS --pmc-out /tmp/acc_$C.json > /dev/null 2>&1)
        done
        rm -rf /tmp/r_pmc_SQ          # third pass: the vector instructions the update issues and its busy cycles (the kernel is VALU-bound: VERDICT round 5)
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d /tmp/r_pmc_SQ -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract --arith $A $BENCH_ARGS --pmc-out /tmp/acc_SQ.json > /dev/null 2>&1)
        rm -rf /tmp/r_pmc_TD /tmp/r_pmc_VM     # fourth and fifth pass: the vector-memory return path (texture-data unit busy / stalled on the L1, L1 accesses) and the memory instruction counts
        if [ "$A" = fast ]; then
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d /tmp/r_pmc_TD -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract --arith $A $BENCH_ARGS --pmc-out /tmp/acc_TD.json > /dev/null 2>&1)
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM -d /tmp/r_pmc_VM -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract --arith $A $BENCH_ARGS --pmc-out /tmp/acc_VM.json > /dev/null 2>&1)
        fi
        python "$ROOT/tools/pmc_to_json.py" "$(db /tmp/r_pmc_FETCH_SIZE)" "$(db /tmp/r_pmc_WRITE_SIZE)" /tmp/acc_FETCH_SIZE.json "$OUT/pmc_tsdf_update.json" "$OUT/pmc_tsdf_update.md" $A "${PMC_CAL:-$ROOT/profiles/r06_pmc_calibration.json}" "$(db /tmp/r_pmc_SQ)" "$(db /tmp/r_pmc_TD)" "$(db /tmp/r_pmc_VM)"
      done ;;
    pltrace)    # the frame loop's own per-frame trace (BF_PIPELINE_TRACE): host enqueue / wait times and the GPU times of detection end, chain begin / end, untraced otherwise
      rm -f "$OUT/pltrace.txt"
      (cd "$ROOT" && BF_PIPELINE_TRACE="$OUT/pltrace.txt" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --no-sweep --long-stream 0 $BENCH_ARGS > "$OUT/bench_pltrace.json" 2> /dev/null; cut -c1-120 "$OUT/bench_pltrace.json"; tail -26 "$OUT/pltrace.txt") ;;
    determinism)
      (cd "$ROOT" && timeout 300 python tools/determinism_check.py ${DET_RUNS:-3} 2>&1 | grep -v amdgpu.ids | tee "$OUT/determinism.txt" | tail -8) ;;
    firstrun)    # one frame loop per process (tools/first_run_check.py) with allocations poisoned: POISONS="*=12345678;sift.hip=12345678;siftmgr.hip:812=ffffffff" (BF_DEBUG_POISON values)
      rm -f "$OUT/first_run.txt"
      IFS=';' read -ra PV <<< "${POISONS:-none;*=12345678;*=0;*=ffffffff}"
      for P in "${PV[@]}"; do
        (cd "$ROOT" && if [ "$P" = none ]; then unset BF_DEBUG_POISON; else export BF_DEBUG_POISON="$P"; fi; timeout 120 python tools/first_run_check.py none 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s|^|$P  |" >> "$OUT/first_run.txt")
      done
      cut -c1-150 "$OUT/first_run.txt" ;;
    racehunt)    # how often does a fresh process produce another trajectory?  RH_ENVS="X=0;BF_PIPELINE_LOOKAHEAD=0" x RH_N processes each
      IFS=';' read -ra RV <<< "${RH_ENVS:-X=0}"
      for V in "${RV[@]}"; do
        TAG=$(echo "$V" | tr ' =/' '___' | tail -c 80)
        rm -f "$OUT/race_$TAG.txt"
        for i in $(seq 1 ${RH_N:-8}); do
          (cd "$ROOT" && env $V timeout 120 python tools/first_run_check.py ${RH_PATTERN:-12345678} 2>&1 | grep -v amdgpu.ids | tail -1 >> "$OUT/race_$TAG.txt")
        done
        echo "== $V: $(grep -c integrated "$OUT/race_$TAG.txt") runs, trajectories: $(grep -o '"integrated": "[0-9a-f]*"' "$OUT/race_$TAG.txt" | sort | uniq -c | sort -rn | awk '{printf "%s x%s  ", substr($3,2,8), $1}')"
      done ;;
    det_bisect)  # the fast batched configuration only, under environment variants (ENVS="A=1;B=2"): which overlap a run-to-run difference needs
      IFS=";" read -ra VARIANTS <<< "${DET_ENVS:-X=0}"
      for V in "${VARIANTS[@]}"; do
        TAG=$(echo "$V" | tr ' =/' '___' | tail -c 80)
        (cd "$ROOT" && env $V timeout 200 python tools/determinism_check.py ${DET_RUNS:-3} fast-batched 2>&1 | grep -v amdgpu.ids > "$OUT/det_$TAG.txt"; echo "== $V"; grep -E "^  block|differing|stale-read" "$OUT/det_$TAG.txt" | cut -c1-300 | head -14; tail -1 "$OUT/det_$TAG.txt")
      done ;;
    hiptrace)   # host side: HIP API calls per thread (totals) and a merged API + kernel window (no counters: --pmc must not be combined with the hip trace)
      rm -rf /tmp/r_hip
      (cd /tmp && timeout 400 rocprofv3 --hip-trace --kernel-trace -d /tmp/r_hip -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --steps 20 --warmup 5 --one-contract $BENCH_ARGS > "$OUT/bench_hiptraced.json" 2> /dev/null)
      D=$(db /tmp/r_hip)
      python "$ROOT/tools/rocpd_hip_trace.py" "$D" > "$OUT/hip_api_totals.txt" 2>&1; head -40 "$OUT/hip_api_totals.txt"
      python "$ROOT/tools/rocpd_hip_trace.py" "$D" --window ${HIP_WINDOW_MS:-5} --end ${HIP_END_MS:-9} > "$OUT/hip_window.txt" 2>&1; wc -l "$OUT/hip_window.txt" ;;
    calibrate)   # FETCH_SIZE / WRITE_SIZE against KNOWN byte counts on the update's own access widths (tools/pmc_calibrate.py)
      for C in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/r_cal_$C
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_cal_$C -o run -- python "$ROOT/tools/pmc_calibrate.py" run > /dev/null 2>&1)
      done
      python "$ROOT/tools/pmc_calibrate.py" read "$(db /tmp/r_cal_FETCH_SIZE)" "$(db /tmp/r_cal_WRITE_SIZE)" "$OUT/pmc_calibration.json" | grep -E "factor|bytes_per_launch" ;;
    sq)   # where do the voxel-update waves spend their cycles: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)
      rm -rf /tmp/r_sq
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
          -d /tmp/r_sq -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 > /dev/null 2>&1)
      python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_sq)" update | tee "$OUT/pmc_sq.txt" | tail -18 ;;
    stream)
      (cd "$ROOT" && timeout 500 python tools/run_sequence.py --frames 5000 --bob 0.3 --voxel 0.004 --buckets 4000000 --blocks 3000000 --tail 35 2>&1 \
          | grep -E "frames|integrated|optimized|counters|allocated|rror" | tee "$OUT/stream5000.txt") ;;
    sens)
      (cd "$ROOT" && timeout 60 python tools/make_sens.py /tmp/r.sens --frames 200 --jpeg 92 | tail -1 && \
          timeout 60 python tools/run_sens.py /tmp/r.sens --voxel 0.004 --buckets 1000000 --blocks 600000 --tail 5 2>&1 | grep -v amdgpu.ids | tee "$OUT/sens200.txt") ;;
    sweep)   # the volume operators alone (tools/tsdf_sweep.py): plain run, then PMC passes (each its own run, --kernel-trace only)
      (cd "$ROOT" && timeout 200 python tools/tsdf_sweep.py $SWEEP_ARGS 2>/dev/null | tee "$OUT/sweep.json" | cut -c1-400)
      (cd "$ROOT" && timeout 200 python tools/tsdf_sweep.py $SWEEP_ARGS --separate 2>/dev/null | tee "$OUT/sweep_separate.json" | cut -c1-400) ;;
    sweep_pmc)
      i=0
      for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
               "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
               "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum" \
               "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1)); rm -rf /tmp/r_sw
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" $SWEEP_ARGS > /dev/null 2>&1)
        python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_sw)" "${PMC_FILTER:-update}" | grep '^|' > "$OUT/sweep_pmc_pass$i.txt"; tail -3 "$OUT/sweep_pmc_pass$i.txt" | cut -c1-200
      done ;;
    bench_pmc_extra)   # the batched update's memory-pipeline counters in the bench configuration (own passes, --kernel-trace only): texture-data busy / stalls, L1 accesses, VMEM instruction counts, L2 hits
      i=0
      for C in "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
               "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
        i=$((i+1)); rm -rf /tmp/r_bx
        (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_bx -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-sweep --long-stream 0 --no-class-surface --steps 20 --warmup 5 --one-contract $BENCH_ARGS > /dev/null 2>&1)
        python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_bx)" "k_update_batch" | grep '^|' > "$OUT/bench_pmc_extra_pass$i.txt"; cat "$OUT/bench_pmc_extra_pass$i.txt" | cut -c1-220
      done ;;
    clocks)
      (rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp" | head -12 | tee -a "$OUT/clocks.txt") ;;
    sweep_trace)
      rm -rf /tmp/r_swt
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_swt -o run -- python "$ROOT/tools/tsdf_sweep.py" $SWEEP_ARGS > /dev/null 2>&1)
      python "$ROOT/tools/rocpd_stats.py" "$(db /tmp/r_swt)" "$OUT/sweep_kernel_stats.md" | head -16 ;;
    *) echo "unknown step $step" ;;
  esac
done
