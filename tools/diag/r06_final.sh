#!/usr/bin/env bash
# round 6, the measurements of record on the shipped build: driver's bench line, kernel trace + timeline, loop trace, PMC (FETCH / WRITE / SQ passes, both contracts), calibration, the 2000-frame 1280x960 @2 mm sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06; mkdir -p "$OUT"; cd "$ROOT"
bash tools/gpu_round.sh 06 calibrate 2>&1 | tail -6
cp "$OUT/pmc_calibration.json" "$ROOT/profiles/r06_pmc_calibration.json"
PMC_CAL="$OUT/pmc_calibration.json" bash tools/gpu_round.sh 06 pmc 2>&1 | tail -8
cp "$OUT/pmc_tsdf_update.json" "$ROOT/profiles/r06_pmc_tsdf_update.json"          # the bench line below reads its traffic / instruction counts from here (same box, same build)
bash tools/gpu_round.sh 06 bench_driver 2>&1 | tail -4
bash tools/gpu_round.sh 06 trace 2>&1 | tail -16
bash tools/gpu_round.sh 06 pltrace 2>&1 | tail -28
(cd "$ROOT" && timeout 1500 python tools/tsdf_sweep.py --frames ${SWEEP_FRAMES:-2000} --stride 1 --width 1280 --height 960 --voxel 0.002 --buckets 8000000 --blocks 3000000 --sweeps 1 2> "$OUT/sweep2000.err" | tee "$OUT/sweep_1280x960_2000.json" | cut -c1-900)
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6 | tee "$OUT/clocks.txt"
