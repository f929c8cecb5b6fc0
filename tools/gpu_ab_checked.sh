#!/bin/bash
# usage: tools/gpu_ab_checked.sh <tag> <variant-to-check> <variants> [frames]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export MULUT_NO_BUILD=1
if [ -x build/ubench_valu ]; then timeout -k 10 120 build/ubench_valu DOT2C DOT2 MADI16 MADU16 SDWAB PKSHR PKSUB CVTI16 PKMADSEL PKMAD BFIOR PERM ADD > gpurun_out/${1}_ubench_valu.txt 2>&1; cat gpurun_out/${1}_ubench_valu.txt; fi
timeout -k 10 300 python tools/check_variant.py $2 > gpurun_out/${1}_check.txt 2>&1 || { echo "variant check failed"; tail -5 gpurun_out/${1}_check.txt; exit 1; }
cat gpurun_out/${1}_check.txt
bash tools/gpu_ab.sh $1 "$3" ${4:-8}
