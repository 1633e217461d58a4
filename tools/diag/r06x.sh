#!/usr/bin/env bash
# round 6, GPU call X: k_blur with 1024 / 512 / 256 threads per 32x32 tile (the gradient blocks riding in k_detect's launch in all three); sift + match + pipeline suites on the 1024 build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06x; mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_sift_gpu.py tests/test_match_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee "$OUT/pytest.txt"
V=$ROOT/bundlefusion_amd/lib/variants
ENVS="BF_X=0;BF_LIB_PATH=$V/libbf_hip_blur256.so;BF_LIB_PATH=$V/libbf_hip_blur512.so;BF_X=1;BF_LIB_PATH=$V/libbf_hip_blur256.so BF_X=1;BF_LIB_PATH=$V/libbf_hip_blur512.so BF_X=1;BF_X=2;BF_LIB_PATH=$V/libbf_hip_blur256.so BF_X=2" bash tools/gpu_round.sh 06x bench_env 2>&1 | grep -v amdgpu.ids | tail -12
