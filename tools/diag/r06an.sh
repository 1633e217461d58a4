#!/usr/bin/env bash
# round 6, last GPU call: the ingest's window filters tiled in LDS - the whole GPU suite (the ingest is pinned bit for bit by the pipeline tests), smoke, kernel statistics, the driver's bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
bash tools/gpu_round.sh 06an tests smoke trace bench_driver 2>&1 | grep -v amdgpu.ids | tail -30
grep -n "k_erode\|k_gauss_depth" gpurun_out/r06an/kernel_stats.md | cut -c1-200
