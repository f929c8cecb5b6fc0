#!/bin/bash
# usage: [MULUT_ASM_SRC=mulut_k1|mulut_detail|mulut_kernels] tools/asm_stats.sh <kernel-name-substring> [extra hipcc flags]
#   compiles mulut_kernels.hip with -save-temps into build/asm, prints registers / scratch / spills of every kernel whose mangled
#   name contains the substring, writes the first one's ISA to build/asm/kernel.s and prints its VALU instruction class histogram
R=$(cd "$(dirname "$0")/.." && pwd)
K=$1; shift
D=$R/build/asm; mkdir -p $D
SRC=${MULUT_ASM_SRC:-mulut_kernels}
( cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Wno-pass-failed "$@" -save-temps=obj -c $R/mulut_amd/csrc/$SRC.hip -o $D/k.o 2>&1 | grep -v "reserved registers" | grep -E "error|warning: v|failed" | head -20 )
S=$D/$SRC-hip-amdgcn-amd-amdhsa-gfx950.s
first=""
for N in $(grep -o "^_Z[A-Za-z0-9_]*$K[A-Za-z0-9_]*:" $S | tr -d ':'); do
  [ -z "$first" ] && first=$N
  echo "kernel $N"
  grep -A14 "\.name: *$N\$" $S | grep "private_segment\|vgpr_count\|sgpr_count\|spill_count" | tr -s ' ' | tr '\n' ' '; echo
done
[ -z "$first" ] && { echo "no kernel matches $K"; exit 1; }
awk -v n="$first:" 'index($0,n)==1{p=1} p{print} /^\.Lfunc_end/{if(p){exit}}' $S > $D/kernel.s      # (a kernel may hold several s_endpgm: cut at the function's end label)
echo "ISA of $first: $(wc -l < $D/kernel.s) lines, scratch ops $(grep -c scratch_ $D/kernel.s), v_readlane/writelane $(grep -c 'v_readlane\|v_writelane' $D/kernel.s)"
