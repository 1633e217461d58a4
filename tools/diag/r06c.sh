#!/usr/bin/env bash
# round 6, GPU call C: the packed build with idle issue slots at one place of the projection (BF_VAR_PAD bit 0 in front, 1 depth->rcp, 2 rcp->coordinates, 3 coordinates->conversion)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06c; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
FR=${FRAMES:-420}; RUNS=${RUNS:-4}
run() { tag=$1; shift; env "$@" timeout 600 python tools/verify_stream.py --frames $FR --runs $RUNS --tag $tag --out "$OUT/verify.jsonl" 2>"$OUT/verify_$tag.err" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['tag'], 'events', j['events'], 'voxels', j['differing_voxels'], 'slices', j['slices_hit'], 'quarters', j['lane_quarters_hit'], 'traj', j['distinct_trajectories'], 'w', j['weight_delta_of_the_odd_value'])"; }
for b in ${PADS:-1 2 4 8}; do run packed_pad$b BF_LIB_PATH=$V/libbf_hip_packed_pad$b.so; done
