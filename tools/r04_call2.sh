#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
timeout -k 10 300 python tools/ab_bench.py --variants base,nofixlist --frames 8 --h 270 --w 480 --rounds 9 > $O/r04b_ab_small.jsonl 2> $O/r04b_ab_small.err; echo "ab small rc=$?"
timeout -k 10 300 python tools/ab_bench.py --variants base,nofixlist --frames 8 --rounds 5 > $O/r04b_ab_p1.jsonl 2> $O/r04b_ab_p1.err; echo "ab p1 rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04b_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/r04b_pytest.log
