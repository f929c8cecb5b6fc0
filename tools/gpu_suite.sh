#!/bin/bash
# One GPU call's worth of validation (gpurun -- 'bash tools/gpu_suite.sh <tag>'): the GPU test suite first; only if it passes, the interleaved stage
# timings at two sizes and the default bench line.  Never starts another GPU step after a failed one.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=${1:-suite}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?
tail -3 $O/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc: stopping"; exit $rc; }
timeout -k 10 300 python tools/ab_bench.py --variants base --frames 8 --h 270 --w 480 --rounds 9 > $O/${TAG}_ab_small.jsonl 2> $O/${TAG}_ab_small.err || { echo "ab small failed"; exit 1; }
timeout -k 10 300 python tools/ab_bench.py --variants base --frames 8 --rounds 5 > $O/${TAG}_ab_p1.jsonl 2> $O/${TAG}_ab_p1.err || { echo "ab p1 failed"; exit 1; }
timeout -k 10 600 python bench.py --cpu-crop 0 --skip-strips > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { echo "bench failed"; exit 1; }
cat $O/${TAG}_ab_small.jsonl $O/${TAG}_ab_p1.jsonl
