#!/usr/bin/env bash
# round 6, GPU call AL: dense verify's closing sum walked as (ty, x) pairs instead of nt run-time modulo tests per thread: match / pipeline / golden suites, kernel statistics, the driver's window
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06al; mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py tests/test_golden_ref_gpu.py -m gpu -x -q 2>&1 | grep -v "ROCm version\|Hostname\|Librccl\|RCCL version\|HIP version" | tail -4 | tee "$OUT/pytest.txt"
bash tools/gpu_round.sh 06al trace 2>&1 | grep -v amdgpu.ids | grep "k_filter_dense_verify\|k_verify_traj\|GPU busy" | cut -c1-200
ENVS="BF_X=0;BF_X=1;BF_X=2" bash tools/gpu_round.sh 06al bench_env 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-330
