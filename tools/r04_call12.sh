#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=${1:-r04z}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?
tail -3 $O/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc: stopping"; exit $rc; }
timeout -k 10 900 python tools/fuzz_parity.py --cases 3000 --seed 404 > $O/${TAG}_fuzz_parity.jsonl 2>&1; echo "fuzz rc=$?"
tail -c 600 $O/${TAG}_fuzz_parity.jsonl
timeout -k 10 600 python tools/fuzz_parity.py --cases 1000 --seed 303 > $O/${TAG}_fuzz_parity_303.jsonl 2>&1; echo "fuzz303 rc=$?"
tail -c 300 $O/${TAG}_fuzz_parity_303.jsonl
