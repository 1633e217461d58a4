#!/usr/bin/env bash
# round 6, closing call: smoke, the two streams at length against their oracle fixtures (the 5000-frame test is gated: BF_LONG_TESTS=1), the driver's bench line on the final tree
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06; mkdir -p "$OUT"; cd "$ROOT"
bash tools/gpu_round.sh 06 smoke
(BF_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_pipeline_baseline_gpu.py -q -m gpu -k "stream_2000 or stream_5000" -rP > "$OUT/streams_at_length_full.txt" 2>&1; grep -E "at length vs|passed|failed|^E " "$OUT/streams_at_length_full.txt" | cut -c1-900 | tee "$OUT/streams_at_length.txt")
bash tools/gpu_round.sh 06 bench_driver 2>&1 | tail -3
