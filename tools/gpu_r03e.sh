#!/usr/bin/env bash
# round 3, GPU call E: re-run of the tests that failed in D, driver-style bench, 5000-frame stream (plain + traced with window statistics), solver scaling, PMC traffic per contract
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03e; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
line() { python - "$1" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]; o=j.get("other_contract")
    print(sys.argv[1].split('/')[-1], "fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f traffic %s host %s" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["share_of_step_time"], r["traffic"], j["config"]["host_thread_ms_per_frame"]), ("| other %s fps %.1f launch_us %.1f frac %.3f" % (o["arith"], o["value"], o["roofline"]["avg_launch_us"], o["roofline"]["frac"])) if o else "")
    if "cpu_baseline" in j: print("   cpu_baseline", json.dumps(j["cpu_baseline"])[:700])
except Exception as e: print("bench failed", sys.argv[1], e)
PY
}
t() { name=$1; shift; timeout 420 python -m pytest "$@" -q -s -x 2>&1 | grep -E "fast contract|fast vs exact|loop closure stream|1280x960|passed|failed|Error|assert|rank " | cut -c1-900 > "$OUT/pytest_$name.txt"; echo "== $name"; tail -6 "$OUT/pytest_$name.txt"; }
t fuse tests/test_match_gpu.py -k fuse
t two_rank tests/test_two_rank_gpu.py
t loop tests/test_pipeline_baseline_gpu.py -k loop_closure
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; line "$OUT/bench_driver.json"
timeout 300 python tools/solver_scaling.py 2>/dev/null | tee "$OUT/solver_scaling.md"
export BF_TSDF_ARITH=fast
(timeout 600 python tools/run_sequence.py --frames 5000 --bob 0.3 --voxel 0.004 --buckets 4000000 --blocks 3000000 --tail 35 2>&1 | grep -E "frames|integrated|optimized|counters|allocated|rror|contract" | tee "$OUT/stream5000.txt")
rm -rf /tmp/r_tr; (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/tools/run_sequence.py" --frames 5000 --bob 0.3 --voxel 0.004 --buckets 4000000 --blocks 3000000 --tail 35 2>&1 | grep -E "^frames|frames [0-9]" > "$OUT/stream5000_traced.txt")
python tools/rocpd_stats_window.py "$(db /tmp/r_tr)" 0.2 > "$OUT/stream5000_windows.md" 2>&1; head -24 "$OUT/stream5000_windows.md" | cut -c1-200
unset BF_TSDF_ARITH
bash tools/gpu_round.sh 03e_pmc pmc > /dev/null 2>&1; cat gpurun_out/r03e_pmc/pmc_tsdf_update.md 2>/dev/null
