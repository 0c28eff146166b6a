#!/bin/bash
# SQ counters of the fused stem kernel on one slice of the headline tree: how busy the matrix
# cores are, what the waves wait for (separate rocprofv3 --pmc passes, kernel trace only).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/stem_sq; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only --tree $R/tests/golden/trees/sycamore_m20_native.json"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -- $CMD > $O/p$i.log 2>&1
done
cd $R
python tools/pmc_dump.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 --match stem2_kernel > $O/stem_sq_counters.txt 2>&1
find $O -name "*.db" -delete; find $O -type d -empty -delete
head -60 $O/stem_sq_counters.txt
