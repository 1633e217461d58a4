#!/usr/bin/env bash
# round 6, last GPU call on the final tree: every batched update three times in the running loop (verify_stream), 200 fresh processes, the streams at length against their oracle fixtures
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06_final4; mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python tools/verify_stream.py --frames 420 --runs 4 --tag final_tree --out "$OUT/verify.jsonl" 2>"$OUT/verify.err" | cut -c1-700
timeout 2400 python tools/determinism_processes.py 200 61 "$OUT/processes_final_tree.jsonl" | cut -c1-600
BF_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_pipeline_baseline_gpu.py -m gpu -q -rP -k "stream_2000 or stream_5000" > "$OUT/streams_full.txt" 2>&1; tail -3 "$OUT/streams_full.txt"; grep -E "configs\[2\] at length|5000|identical|ATE" "$OUT/streams_full.txt" | cut -c1-600 | head -12 | tee "$OUT/streams_at_length.txt"
