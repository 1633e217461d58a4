#!/usr/bin/env bash
# round 6, GPU call A: execution-error rate of the batched update per library variant (tools/verify_stream.py) + the per-mechanism reproducers (tools/probe/bf_hazards)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06a; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
FR=${FRAMES:-420}; RUNS=${RUNS:-4}
run() { tag=$1; shift; env "$@" timeout 600 python tools/verify_stream.py --frames $FR --runs $RUNS --tag $tag --out "$OUT/verify.jsonl" 2>"$OUT/verify_$tag.err" | cut -c1-700; }
run product
run packed BF_LIB_PATH=$V/libbf_hip_packed.so
run vload BF_LIB_PATH=$V/libbf_hip_vload.so
run packed_vload BF_LIB_PATH=$V/libbf_hip_packed_vload.so
run product_nooverlap BF_DEBUG_NO_OVERLAP=1
run packed_nooverlap BF_LIB_PATH=$V/libbf_hip_packed.so BF_DEBUG_NO_OVERLAP=1
timeout 400 tools/probe/bf_hazards 6 > "$OUT/hazards.json" 2> "$OUT/hazards.err"; cut -c1-3000 "$OUT/hazards.json"
# the driver's window with R compute units kept off the volume stream
ENVS="BF_VOLUME_CU_RESERVE=0;BF_VOLUME_CU_RESERVE=16;BF_VOLUME_CU_RESERVE=32;BF_VOLUME_CU_RESERVE=64" bash tools/gpu_round.sh 06a bench_env 2>&1 | tail -8
