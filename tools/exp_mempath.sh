#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-mem}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES"
P2="TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY"
P3="TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_TCR_TCP_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE"
P4="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"
for lib in memonly peel; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    CTG_LIB=$R/cotengra_amd/lib/exp/libctg_$lib.so timeout 200 rocprofv3 --kernel-trace --pmc $P -d $O/${lib}_p$i -- python $R/tools/exp_mempath.py > $O/${lib}_p$i.log 2>&1
    python $R/tools/pmc_dump.py $O/${lib}_p$i 2>/dev/null | grep -A12 "stream_kernel\|elementwise" | grep -v "^--" | cut -c1-150 > $O/${lib}_p$i.txt
    rm -rf $O/${lib}_p$i
  done
done
cat $O/*.txt | cut -c1-230
