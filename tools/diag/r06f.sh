#!/usr/bin/env bash
# round 6, GPU call F: the GPU suite on the current tree, then run-to-run identity over fresh processes on the shipped build (200) and on the packed build (100)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06f; mkdir -p "$OUT"; cd "$ROOT"
bash tools/gpu_round.sh 06f tests 2>&1 | tail -30
timeout 2400 python tools/determinism_processes.py ${NPROD:-200} 61 "$OUT/processes_product.jsonl" | cut -c1-600
BF_LIB_PATH=$ROOT/bundlefusion_amd/lib/variants/libbf_hip_packed.so timeout 1500 python tools/determinism_processes.py ${NPACK:-100} 61 "$OUT/processes_packed.jsonl" | cut -c1-600
