#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=${1:-r04r}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?
tail -3 $O/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc: stopping"; exit $rc; }
timeout -k 10 600 python tools/ab_bench.py --variants "$2" --frames 32 --rounds 5 > $O/${TAG}_ab_p1.jsonl 2> $O/${TAG}_ab_p1.err || { echo "ab p1 failed"; exit 1; }
cat $O/${TAG}_ab_p1.jsonl
