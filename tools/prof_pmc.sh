#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (one --pmc set per run, no tracing
# options mixed in), then summarise per kernel.   usage: tools/prof_pmc.sh <tag> [bench args...]
# Run on the GPU box from anywhere; results go to $GRAFT_REPO_ROOT/gpurun_out/pmc_<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-run}; shift || true
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# build once, outside the profiler: a profiled process has the GPU initialised and must not start hipcc
( cd "$R" && python -c 'from mulut_amd import _native; _native.build()' ) || exit 1
export MULUT_NO_BUILD=1
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_MISS_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python "$R/bench.py" --cpu-crop 0 --steps 2 --warmup 1 --skip-other --skip-strips "$@" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
  i=$((i+1))
done
python "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"
