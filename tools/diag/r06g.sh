#!/usr/bin/env bash
# round 6, GPU call G: the current tree (lean update, lagged default, CU reserve): verify stream, GPU suite, schedule / queue variants of the driver's window
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06g; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
timeout 600 python tools/verify_stream.py --frames 420 --runs 4 --tag product_lean --out "$OUT/verify.jsonl" 2>"$OUT/verify.err" | cut -c1-400
ENVS="BF_X=0;BF_PIPELINE_SOLVE_LAG=0;BF_VOLUME_CU_RESERVE=0;BF_VOLUME_CU_RESERVE=64;BF_PIPELINE_PAIR_STREAMS=1;BF_LIB_PATH=$V/libbf_hip_chainprio.so;BF_LIB_PATH=$V/libbf_hip_chainprio.so BF_PIPELINE_PAIR_STREAMS=1;BF_PIPELINE_SOLVE_LAG=0 BF_VOLUME_CU_RESERVE=0" bash tools/gpu_round.sh 06g bench_env 2>&1 | grep -v amdgpu.ids | tail -12
bash tools/gpu_round.sh 06g tests 2>&1 | tail -30
