#!/bin/bash
# usage: tools/gpu_ab.sh <tag> <variants> [frames] [extra ab_bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
export MULUT_NO_BUILD=1
TAG=$1; V=$2; F=${3:-8}; shift 3 2>/dev/null
timeout -k 10 600 python tools/ab_bench.py --variants "$V" --frames $F --rounds 7 "$@" > $O/${TAG}_ab.jsonl 2> $O/${TAG}_ab.err || { echo "ab failed"; tail -3 $O/${TAG}_ab.err; exit 1; }
cat $O/${TAG}_ab.jsonl
