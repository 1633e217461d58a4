#!/usr/bin/env bash
# round 6, GPU call Z: the batched update's workgroups per CU capped by unused dynamic LDS (6 by registers; 5 / 4 / 3 / 2 by 32 / 40 / 52 / 64 KB), with the CU reserve 32 and 0
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06z; mkdir -p "$OUT"; cd "$ROOT"
ENVS="BF_X=0;BF_DEBUG_UPDATE_LDS=32768;BF_DEBUG_UPDATE_LDS=40960;BF_DEBUG_UPDATE_LDS=53248;BF_DEBUG_UPDATE_LDS=65536;BF_DEBUG_UPDATE_LDS=40960 BF_VOLUME_CU_RESERVE=0;BF_DEBUG_UPDATE_LDS=53248 BF_VOLUME_CU_RESERVE=0;BF_X=1" bash tools/gpu_round.sh 06z bench_env 2>&1 | grep -v amdgpu.ids | tail -12
