#!/usr/bin/env bash
# round 6, GPU call H: loop trace + kernel timeline of the default schedule; two-rank / batch tests on the divided batch march; the driver's bench line incl. class surface and the 5000-frame stream
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06h; mkdir -p "$OUT"; cd "$ROOT"
TESTS="tests/test_two_rank_gpu.py tests/test_tsdf_batch_gpu.py tests/test_sift_gpu.py" bash tools/gpu_round.sh 06h tests_sel 2>&1 | tail -6
bash tools/gpu_round.sh 06h pltrace 2>&1 | tail -45
bash tools/gpu_round.sh 06h trace 2>&1 | tail -20
(time bash tools/gpu_round.sh 06h bench_driver) 2>&1 | tail -8
