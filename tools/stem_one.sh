#!/bin/bash
# Kernel development: compile ONE instantiation of the fused stem kernel (seconds) and print its register
# count and the instruction mix of its steady-state loop.
#   tools/stem_one.sh "false,false,1,1,2,1,true,0,false,true,false,false,0,false,true,true" [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
T="$1"; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "-DCTG_STEM_DEV_ONE=$T" "$@" \
    -c "$R/cotengra_amd/csrc/ctg_stem.hip" -o /tmp/stem_one.o -save-temps=obj 2>&1 | grep -v warning | grep -i "error" -A5 | head -20
S=/tmp/ctg_stem-hip-amdgcn-amd-amdhsa-gfx950.s
grep "vgpr_count\|vgpr_spill\|group_segment" $S
python3 "$R/tools/asm_mix.py" $S
