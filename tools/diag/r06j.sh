#!/usr/bin/env bash
# round 6, GPU call J: the matching chain on compute units of its own (BF_CHAIN_CU_EXCLUSIVE=R), everything else on the rest
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06j; mkdir -p "$OUT"; cd "$ROOT"
ENVS="BF_X=0;BF_CHAIN_CU_EXCLUSIVE=16;BF_CHAIN_CU_EXCLUSIVE=32;BF_CHAIN_CU_EXCLUSIVE=48;BF_CHAIN_CU_EXCLUSIVE=64;BF_CHAIN_CU_EXCLUSIVE=32 BF_PIPELINE_SOLVE_LAG=0;BF_CHAIN_CU_EXCLUSIVE=32 BF_PIPELINE_DEPTH=3" bash tools/gpu_round.sh 06j bench_env 2>&1 | grep -v amdgpu.ids | tail -12
