#!/usr/bin/env bash
# round 6, GPU call D: the packed build with ONE stage of the projection as scalar instructions (BF_VAR_SCALAR_PROJECT 1 depth, 2 numerators, 4 image coordinates)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06d; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
FR=${FRAMES:-420}; RUNS=${RUNS:-4}
run() { tag=$1; shift; env "$@" timeout 600 python tools/verify_stream.py --frames $FR --runs $RUNS --tag $tag --out "$OUT/verify.jsonl" 2>"$OUT/verify_$tag.err" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['tag'], 'events', j['events'], 'voxels', j['differing_voxels'], 'slices', j['slices_hit'], 'quarters', j['lane_quarters_hit'], 'traj', j['distinct_trajectories'], 'w', j['weight_delta_of_the_odd_value'])"; }
for b in 1 2 4; do run packed_sp$b BF_LIB_PATH=$V/libbf_hip_packed_sp$b.so; done
