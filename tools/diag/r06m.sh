#!/usr/bin/env bash
# round 6, GPU call M: which streams share a hardware queue is decided by their creation order - try the orders that separate the two detection streams
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06m; mkdir -p "$OUT"; cd "$ROOT"
ENVS="BF_PIPELINE_STREAM_ORDER=DSIPQ;BF_PIPELINE_STREAM_ORDER=DSPIQ;BF_PIPELINE_STREAM_ORDER=DIPSQ;BF_PIPELINE_STREAM_ORDER=DPISQ;BF_PIPELINE_STREAM_ORDER=DIPQS;BF_PIPELINE_STREAM_ORDER=DPSIQ;BF_PIPELINE_STREAM_ORDER=DPQIS;BF_PIPELINE_STREAM_ORDER=SDIPQ" bash tools/gpu_round.sh 06m bench_env 2>&1 | grep -v amdgpu.ids | tail -12
