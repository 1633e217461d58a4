#!/usr/bin/env bash
# round 6, GPU call B: literal instruction window of the packed build in isolation + source-perturbed packed variants in the frame loop
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06b; mkdir -p "$OUT"; cd "$ROOT"
V=$ROOT/bundlefusion_amd/lib/variants
timeout 300 tools/probe/bf_hazards 5 window > "$OUT/hazards_window.json" 2> "$OUT/hazards.err"; cut -c1-2500 "$OUT/hazards_window.json"
FR=${FRAMES:-420}; RUNS=${RUNS:-4}
run() { tag=$1; shift; env "$@" timeout 600 python tools/verify_stream.py --frames $FR --runs $RUNS --tag $tag --out "$OUT/verify.jsonl" 2>"$OUT/verify_$tag.err" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['tag'], 'events', j['events'], 'voxels', j['differing_voxels'], 'slices', j['slices_hit'], 'quarters', j['lane_quarters_hit'], 'traj', j['distinct_trajectories'], 'w', j['weight_delta_of_the_odd_value'])"; }
run packed BF_LIB_PATH=$V/libbf_hip_packed.so
run packed_scalarproj BF_LIB_PATH=$V/libbf_hip_packed_scalarproj.so
run packed_reverse BF_LIB_PATH=$V/libbf_hip_packed_reverse.so
run product
