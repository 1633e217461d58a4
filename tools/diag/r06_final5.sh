#!/usr/bin/env bash
# round 6: the PMC file re-collected on the final tree with the memory-pipeline passes (TD / TCP / VMEM) added, then the driver's bench line reading it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06_final5; mkdir -p "$OUT"; cd "$ROOT"
rm -f "$OUT/pmc_tsdf_update.json" "$OUT/pmc_tsdf_update.md"
bash tools/gpu_round.sh 06_final5 pmc 2>&1 | grep -v amdgpu.ids | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_final5/pmc_tsdf_update.json"))
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["fast"].get("mem_pipe", {}).items() if k not in ("per_launch", "note")})
PY
cp "$OUT/pmc_tsdf_update.json" profiles/r06_pmc_tsdf_update.json
bash tools/gpu_round.sh 06_final5 bench_driver 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final5/bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["mem_pipe"], d["roofline"]["valu"]["frac"], d["roofline"]["traffic"])
PY
