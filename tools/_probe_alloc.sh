export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r02i; mkdir -p $OUT
db() { ls -S "$1"/*/*_results.db "$1"/*_results.db 2>/dev/null | head -1; }
cd /tmp
# 1. no-overlap trace: intrinsic kernel durations
rm -rf /tmp/t1; timeout 300 rocprofv3 --kernel-trace -d /tmp/t1 -o run -- python $ROOT/tools/tsdf_sweep.py --no-overlap > /dev/null 2>&1
python $ROOT/tools/rocpd_stats.py "$(db /tmp/t1)" $OUT/sweep_nooverlap_stats.md | head -12
# 2. SQ counters of the alloc kernels with overlap
rm -rf /tmp/t2; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/t2 -o run -- python $ROOT/tools/tsdf_sweep.py > /dev/null 2>&1
python $ROOT/tools/rocpd_pmc.py "$(db /tmp/t2)" alloc | grep '^|' | tee $OUT/alloc_sq.txt | cut -c1-160
rm -rf /tmp/t3; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d /tmp/t3 -o run -- python $ROOT/tools/tsdf_sweep.py > /dev/null 2>&1
python $ROOT/tools/rocpd_pmc.py "$(db /tmp/t3)" alloc | grep '^|' | tee $OUT/alloc_sq2.txt | cut -c1-160
