#!/usr/bin/env bash
# round 6, GPU call K: the lagged solve's stream on the reserved compute units (own hardware queue)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06k; mkdir -p "$OUT"; cd "$ROOT"
ENVS="BF_X=0;BF_SOLVE_CUS=32;BF_SOLVE_CUS=64 BF_VOLUME_CU_RESERVE=64;BF_SOLVE_CUS=16;BF_SOLVE_CUS=256;BF_SOLVE_CUS=32 BF_PIPELINE_OWN_QUEUES=2" bash tools/gpu_round.sh 06k bench_env 2>&1 | grep -v amdgpu.ids | tail -12
BF_SOLVE_CUS=32 bash tools/gpu_round.sh 06k pltrace 2>&1 | tail -32
