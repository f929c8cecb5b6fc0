#!/bin/bash
# first GPU check of a kernel change: parity of the kernel variants on small frames, then an interleaved A/B timing
#   usage (GPU box): bash tools/gpu_quick.sh "<ab_bench variant list>" [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=${1:-base,base+tube_pipelined=0}
TAG=${2:-quick}
OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
cd "$R"
export MULUT_NO_BUILD=1
timeout -k 10 300 python tools/check_variants.py > "$OUT/variants.log" 2>&1; echo "check_variants rc=$?" | tee -a "$OUT/summary.txt"
grep -c OK "$OUT/variants.log" | sed 's/^/  OK lines: /' | tee -a "$OUT/summary.txt"
grep MISMATCH "$OUT/variants.log" | head -5 | tee -a "$OUT/summary.txt"
timeout -k 10 400 python tools/ab_bench.py --variants "$V" --frames 8 --rounds 5 > "$OUT/ab.log" 2>&1; echo "ab_bench rc=$?" | tee -a "$OUT/summary.txt"
grep '^{' "$OUT/ab.log" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/ab.log" | grep -v '^{' 
