#!/usr/bin/env bash
# round 6, GPU call Y: the driver's window with / without the volume operators beside the loop (tools/loop_parts.py)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06y; mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python tools/loop_parts.py 3 2> "$OUT/loop_parts.err" | tee "$OUT/loop_parts.jsonl"
tail -3 "$OUT/loop_parts.err"
