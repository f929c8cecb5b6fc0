#!/bin/bash
# Extra SQ counter passes (LDS latency, instruction fetch, issue stalls) for bench.py; see prof_pmc.sh.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-extra}; shift || true
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# build once, outside the profiler: a profiled process has the GPU initialised and must not start hipcc
( cd "$R" && python -c 'from mulut_amd import _native; _native.build()' ) || exit 1
export MULUT_NO_BUILD=1
rocprofv3 -L > "$OUT/counters.txt" 2>&1
PASSES=(
 "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU"
 "SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
 "SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python "$R/bench.py" --cpu-crop 0 --steps 2 --warmup 1 --skip-other --skip-strips "$@" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
  i=$((i+1))
done
python "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json" && python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-28s %.4g" % (c, x["mean_per_dispatch"]))
PY
