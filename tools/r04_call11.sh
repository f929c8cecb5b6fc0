#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=${1:-r04w}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?
tail -3 $O/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc: stopping"; exit $rc; }
timeout -k 10 300 python tools/prof_phases.py > $O/${TAG}_tube2_phases.json 2> $O/${TAG}_phases.err || { echo "phases failed"; exit 1; }
python - <<PY
import json
r=json.load(open("$O/${TAG}_tube2_phases.json"))
print(r["in_kernel_clock_ghz"], r["wave_lifetime_us"], [(p["phase"][:20], p["share"], p["cycles_per_tile_and_wave"]) for p in r["phases"]])
PY
timeout -k 10 600 python tools/ab_bench.py --variants base --frames 32 --rounds 5 > $O/${TAG}_ab_p1.jsonl 2> $O/${TAG}_ab_p1.err || { echo "ab p1 failed"; exit 1; }
cat $O/${TAG}_ab_p1.jsonl
