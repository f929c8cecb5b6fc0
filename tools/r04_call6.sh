#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=${1:-r04f}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?
tail -3 $O/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc: stopping"; exit $rc; }
timeout -k 10 300 python tools/prof_k1.py > $O/${TAG}_k1_phases.txt 2> $O/${TAG}_k1_phases.err || { echo "prof_k1 failed"; exit 1; }
cat $O/${TAG}_k1_phases.txt
timeout -k 10 300 python tools/ab_bench.py --variants base --frames 8 --h 270 --w 480 --rounds 9 > $O/${TAG}_ab_small.jsonl 2> $O/${TAG}_ab_small.err || { echo "ab small failed"; exit 1; }
timeout -k 10 300 python tools/ab_bench.py --variants base --frames 8 --rounds 5 > $O/${TAG}_ab_p1.jsonl 2> $O/${TAG}_ab_p1.err || { echo "ab p1 failed"; exit 1; }
timeout -k 10 600 python bench.py --cpu-crop 0 --skip-strips > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { echo "bench failed"; exit 1; }
cat $O/${TAG}_ab_small.jsonl $O/${TAG}_ab_p1.jsonl
