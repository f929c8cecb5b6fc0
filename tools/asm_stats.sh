#!/bin/bash
# usage: tools/asm_stats.sh <kernel-name-substring> [extra hipcc flags]   -> VGPRs, scratch, instruction histogram
R=$(cd "$(dirname "$0")/.." && pwd)
K=$1; shift
D=$R/build/asm; mkdir -p $D
( cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -save-temps=obj -c $R/mulut_amd/csrc/mulut_kernels.hip -o k.o 2>&1 | grep -E "error" )
S=$D/mulut_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
N=$(grep -o "^_Z[A-Za-z0-9_]*$K[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
echo "kernel $N"
grep -A12 "\.name: *$N\$" $S | grep "private_segment\|vgpr_count\|sgpr_count"
awk -v n="$N:" 'index($0,n)==1{p=1} p{print} /s_endpgm/{if(p){exit}}' $S > $D/kernel.s
echo "lines $(wc -l < $D/kernel.s)  scratch ops $(grep -c scratch_ $D/kernel.s)"
