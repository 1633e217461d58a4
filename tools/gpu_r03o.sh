#!/usr/bin/env bash
# round 3, GPU call O: the round's standard measurements on the FINAL code (texel gathers): full GPU suite, smoke, PMC traffic of the fast contract, driver-style bench, trace, 5000-frame stream
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03o; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
(timeout 900 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -14 | tee "$OUT/pytest_gpu.txt")
(timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee "$OUT/smoke.txt")
PMC_CONTRACTS=fast bash tools/gpu_round.sh 03o_pmc pmc > /dev/null 2>&1; cat gpurun_out/r03o_pmc/pmc_tsdf_update.md 2>/dev/null
python - <<'PY'
import json
try:
    new=json.load(open('gpurun_out/r03o_pmc/pmc_tsdf_update.json')); old=json.load(open('profiles/r03_pmc_tsdf_update.json')); old.update(new); json.dump(old, open('profiles/r03_pmc_tsdf_update.json','w'), indent=1); print('pmc json updated', {k: round(old[k]['fused']['hbm_bytes_per_visited_block']) for k in old})
except Exception as e: print('pmc update failed', e)
PY
cp profiles/r03_pmc_tsdf_update.json "$OUT/r03_pmc_tsdf_update.json"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; python - "$OUT/bench_driver.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; o=j.get("other_contract")
print("bench fps %.1f ms/step %.3f launch_us %.1f frac %.3f traffic %.0f hbm_frac %.3f share %.2f | other %s fps %.1f launch_us %.1f frac %.3f" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["traffic"] or 0, r["hbm_frac_measured"] or 0, r["share_of_step_time"], o["arith"], o["value"], o["roofline"]["avg_launch_us"], o["roofline"]["frac"]))
print("host", j["config"]["host_thread_ms_per_frame"]); print("cpu_baseline", json.dumps(j["cpu_baseline"])[:300])
PY
timeout 300 python bench.py --no-cpu-baseline --one-contract > "$OUT/bench_default.json" 2> /dev/null; python -c "
import json; j=json.load(open('$OUT/bench_default.json')); r=j['roofline']; print('bench default (200 steps) fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"
rm -rf /tmp/r_tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/fast_traced.json" 2>/dev/null)
python tools/rocpd_stats.py "$(db /tmp/r_tr)" "$OUT/kernel_stats.md" --exclude "Cijk_,at::native" | head -7
python tools/rocpd_timeline.py "$(db /tmp/r_tr)" 0.5 "Cijk_,at::native" > "$OUT/timeline.txt" 2>&1; grep -E "k_update_apx -> void|^queue|GPU busy" "$OUT/timeline.txt" | cut -c1-200
(BF_TSDF_ARITH=fast timeout 400 python tools/run_sequence.py --frames 5000 --bob 0.3 --voxel 0.004 --buckets 4000000 --blocks 3000000 --tail 35 2>&1 | grep -E "frames|integrated|optimized|counters|allocated|rror|contract" | tee "$OUT/stream5000.txt")
