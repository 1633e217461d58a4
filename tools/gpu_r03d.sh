#!/usr/bin/env bash
# round 3, GPU call D: parity tests of the round (each under its own timeout), the new bench line, list buffers x4 (gaps), CPU placement of the host threads vs the two modes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03d; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
line() { python - "$1" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]; o=j.get("other_contract")
    print(sys.argv[1].split('/')[-1], "fps %.1f ms/step %.3f launch_us %.1f frac %.3f share %.2f host %s" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], r["share_of_step_time"], j["config"]["host_thread_ms_per_frame"]), ("| other %s fps %.1f launch_us %.1f frac %.3f" % (o["arith"], o["value"], o["roofline"]["avg_launch_us"], o["roofline"]["frac"])) if o else "")
    if "cpu_baseline" in j: print("   cpu_baseline", json.dumps(j["cpu_baseline"])[:900])
except Exception as e: print("bench failed", sys.argv[1], e)
PY
}
t() { name=$1; shift; timeout 420 python -m pytest "$@" -q -s -x 2>&1 | grep -E "fast contract|fast vs exact|loop closure stream|1280x960|passed|failed|Error|assert|rank " | cut -c1-900 > "$OUT/pytest_$name.txt"; echo "== $name"; tail -6 "$OUT/pytest_$name.txt"; }
(nproc; lscpu | grep -iE "numa|model name|socket|thread"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showtoponuma 2>/dev/null | grep -i numa) > "$OUT/cpu_topology.txt" 2>&1; cat "$OUT/cpu_topology.txt" | head -20
t tsdf_fast tests/test_tsdf_fast_gpu.py
t tsdf tests/test_tsdf_gpu.py
t fuse tests/test_match_gpu.py -k fuse
t two_rank tests/test_two_rank_gpu.py
t loop tests/test_pipeline_baseline_gpu.py -k loop_closure
t sweep2mm tests/test_pipeline_baseline_gpu.py -k 1280x960
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; line "$OUT/bench_driver.json"; tail -2 "$OUT/bench_driver.err"
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_fast_$i.json" 2>/dev/null; line "$OUT/bench_fast_$i.json"; done
NC=$(nproc)
for cpus in "0-15" "$((NC/2))-$((NC/2+15))" "$((NC-16))-$((NC-1))"; do
  for a in exact; do taskset -c $cpus timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract --arith $a > "$OUT/bench_${a}_cpus_$cpus.json" 2>/dev/null; line "$OUT/bench_${a}_cpus_$cpus.json"; done
done
rm -rf /tmp/r_tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/r_tr -o run -- python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/fast_traced.json" 2>/dev/null)
python tools/rocpd_stats.py "$(db /tmp/r_tr)" "$OUT/fast_kernel_stats.md" --exclude "Cijk_,at::native" | head -8
python tools/rocpd_timeline.py "$(db /tmp/r_tr)" 0.5 "Cijk_,at::native" > "$OUT/fast_timeline.txt" 2>&1; grep -E "k_update_apx -> void|^queue" "$OUT/fast_timeline.txt" | cut -c1-200; line "$OUT/fast_traced.json"
