#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
TAG=${1:-r04e}
python -c 'from mulut_amd import _native; _native.build()' || exit 1
export MULUT_NO_BUILD=1
timeout -k 10 300 python tools/prof_k1.py > $O/${TAG}_k1_phases.txt 2> $O/${TAG}_k1_phases.err || { echo "prof_k1 failed"; tail -5 $O/${TAG}_k1_phases.err; exit 1; }
cat $O/${TAG}_k1_phases.txt
bash tools/prof_round.sh $TAG > $O/prof_${TAG}.log 2>&1 || { echo "prof_round failed"; exit 1; }
tail -2 $O/prof_${TAG}.log
cp profiles/kernel_counters.json profiles/hbm_traffic.json $O/ 2>/dev/null
echo done
