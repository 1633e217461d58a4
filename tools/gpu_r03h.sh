#!/usr/bin/env bash
# round 3, GPU call H: the round's standard measurements on the final code: full GPU suite, smoke, driver-style bench (both contracts + cpu baseline), trace, solver scaling
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03h; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
(timeout 1200 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -30 | tee "$OUT/pytest_gpu.txt")
(timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee "$OUT/smoke.txt")
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; python - "$OUT/bench_driver.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; o=j.get("other_contract")
print("bench fps %.1f ms/step %.3f launch_us %.1f frac %.3f traffic %.0f hbm_frac %.3f share %.2f | other %s fps %.1f launch_us %.1f frac %.3f" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["traffic"] or 0, r["hbm_frac_measured"] or 0, r["share_of_step_time"], o["arith"], o["value"], o["roofline"]["avg_launch_us"], o["roofline"]["frac"]))
print("cpu_baseline", json.dumps(j["cpu_baseline"])[:400])
PY
timeout 300 python bench.py > "$OUT/bench_default.json" 2> /dev/null; python -c "
import json; j=json.load(open('$OUT/bench_default.json')); r=j['roofline']; print('bench default (200 steps) fps %.1f launch_us %.1f frac %.3f share %.2f' % (j['value'], r['avg_launch_us'], r['frac'], r['share_of_step_time']))"
rm -rf /tmp/r_tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/fast_traced.json" 2>/dev/null)
python tools/rocpd_stats.py "$(db /tmp/r_tr)" "$OUT/kernel_stats.md" --exclude "Cijk_,at::native" | head -8
python tools/rocpd_timeline.py "$(db /tmp/r_tr)" 0.5 "Cijk_,at::native" > "$OUT/timeline.txt" 2>&1; grep -E "k_update_apx -> void|^queue|GPU busy" "$OUT/timeline.txt" | cut -c1-200
timeout 400 python tools/solver_scaling.py --n 500 1000 1500 2000 2> /dev/null | tee "$OUT/solver_scaling.md"
