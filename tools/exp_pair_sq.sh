#!/bin/bash
# SQ counters of the 16-bit tiled kernels (pair_mfma_h2_kernel / pair_mfma_bf3_kernel) on one GEMM-like step:
# tools/exp_pair_sq.sh [R K N]   (separate rocprofv3 --pmc passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pair_sq; rm -rf $O; mkdir -p $O
RR=${1:-65536}; KK=${2:-512}; NN=${3:-512}
CMD="python $R/tools/bench_pair.py ak,kb->ab a=$RR,k=$KK,b=$NN 3"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -- $CMD > $O/p$i.log 2>&1
done
cd $R
python tools/pmc_dump.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 --match pair_mfma_ > $O/pair_sq_counters.txt 2>&1
find $O -name "*.db" -delete; find $O -type d -empty -delete
head -60 $O/pair_sq_counters.txt
