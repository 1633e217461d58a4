#!/usr/bin/env bash
# bisect on one box: the driver's bench of four earlier commits of this round and of the working tree
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04i; mkdir -p "$OUT"; export TMPDIR=/tmp
show() { python -c "
import json,sys; j=json.load(open('$1')); r=j['roofline']; h=j['config']['host_thread_ms_per_frame']
print('$2 fps %.1f ms/step %.3f launch_us %.1f host %s' % (j['value'], j['ms_per_step'], r['avg_launch_us'], h))"; }
for c in db463c4 1d413ce 3439327 5288a36; do
  (cd "$ROOT/gpurun_ab/wt_$c" && BF_PIPELINE_DEPTH=2 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || tail -3 "$OUT/bench_$c.err")
  show "$OUT/bench_$c.json" $c
done
cd "$ROOT"
BF_PIPELINE_DEPTH=2 BF_SCENE_SPLIT_PREP=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_head_s0.json" 2> "$OUT/bench_head_s0.err"; show "$OUT/bench_head_s0.json" head_s0
(cd "$ROOT/gpurun_ab/wt_db463c4" && timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-contract > "$OUT/bench_db463c4_b.json" 2>/dev/null); show "$OUT/bench_db463c4_b.json" db463c4_again
