#!/usr/bin/env bash
# round 6, last call: the whole GPU suite and the driver's bench line on the final tree
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
bash tools/gpu_round.sh 06 tests 2>&1 | tail -24
bash tools/gpu_round.sh 06 smoke
bash tools/gpu_round.sh 06 bench_driver 2>&1 | tail -3
