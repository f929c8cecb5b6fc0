#!/bin/bash
# One round's evidence for bench.py's workload: rocprofv3 kernel stats + four PMC passes (never mixed with tracing), then
# profiles/kernel_counters.json and profiles/hbm_traffic.json tied to the source hash of the kernels.
#   usage (GPU box): bash tools/prof_round.sh r02        -> gpurun_out/prof_r02/ ; copy what is to be judged into profiles/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
OUT=$R/gpurun_out/prof_$TAG; mkdir -p "$OUT"
( cd "$R" && python -c 'from mulut_amd import _native; _native.build()' ) || exit 1
export MULUT_NO_BUILD=1          # a profiled process has the GPU initialised and must not start hipcc
cd /tmp && export TMPDIR=/tmp
# DIST=real|noise profiles that input distribution; the JSONs then carry the suffix (profiles/kernel_counters_real.json ...)
DIST=${DIST:-natural}
ARGS="--cpu-crop 0 --steps 4 --warmup 2 --skip-other --skip-strips --dist $DIST"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$R/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || tail -3 "$OUT/stats.log"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
grep '^{' "$OUT/stats.log" | tail -1 > "$OUT/bench_under_stats.json"
declare -A P
P[sq]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P[lds]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P[rd]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum"
P[wr]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_MISS_sum"
for k in sq lds rd wr; do
  timeout -k 10 300 rocprofv3 --pmc ${P[$k]} --output-format csv -d "$OUT/$k" -- python "$R/bench.py" $ARGS > "$OUT/$k.log" 2>&1 || { echo "pass $k failed"; tail -3 "$OUT/$k.log"; }
done
python "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json"
if [ "$DIST" = natural ]; then python "$R/tools/make_profile_json.py" "$OUT" "$TAG"; else python "$R/tools/make_profile_json.py" "$OUT" "$TAG" "_$DIST"; fi
