#!/usr/bin/env bash
# round 3, GPU call L: is the fast voxel update bound by the texture-address / L1 path of its divergent gathers?  (TA / TCP / TD counters around the sweep)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03l; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
(cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ)_[A-Za-z0-9_]+" | sort -u > "$OUT/counters_available.txt"); wc -l "$OUT/counters_available.txt"
export BF_TSDF_ARITH=fast
i=0
for C in "TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/r_sw
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/r_sw -o run -- python "$ROOT/tools/tsdf_sweep.py" > "$OUT/pass$i.log" 2>&1)
  python "$ROOT/tools/rocpd_pmc.py" "$(db /tmp/r_sw)" "apx<2" 2>/dev/null | grep '^|' > "$OUT/ta_pass$i.txt"; cat "$OUT/ta_pass$i.txt" | awk -F'|' '{printf "%s | %s | %s\n",$3,$4,$5}'; grep -iE "error|invalid|not found" "$OUT/pass$i.log" | head -3
done
