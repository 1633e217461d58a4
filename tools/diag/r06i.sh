#!/usr/bin/env bash
# round 6, GPU call I: hardware-queue aliasing of the solve / ingest / second detection streams (the lagged schedule's wait moved into the ingest event)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06i; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
ENVS="BF_X=0;GPU_MAX_HW_QUEUES=8;BF_PIPELINE_OWN_QUEUES=1;BF_PIPELINE_OWN_QUEUES=3;BF_PIPELINE_OWN_QUEUES=11;BF_PIPELINE_OWN_QUEUES=15;BF_PIPELINE_OWN_QUEUES=3 BF_LIB_PATH=$V/libbf_hip_chainprio.so;GPU_MAX_HW_QUEUES=8 BF_PIPELINE_OWN_QUEUES=3" bash tools/gpu_round.sh 06i bench_env 2>&1 | grep -v amdgpu.ids | tail -12
