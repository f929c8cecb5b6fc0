#!/bin/bash
# Quick rocprofv3 passes over the C++ C-ABI harness (tools/cabi_bench.cpp: starts in a second, no Python).
#   usage: tools/prof_quick.sh <tag> "<cabi_bench args>" [stats|sq|lds|mem|ta|ta2 ...]
# stats = --kernel-trace --stats; the others are one --pmc pass each (never mixed with tracing).  Results under
# gpurun_out/pq_<tag>/ ; a one-line-per-counter summary is printed.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
OUT=$R/gpurun_out/pq_$TAG; mkdir -p "$OUT"
BIN=$R/build/tools/cabi_bench
[ -x "$BIN" ] || { echo "build $BIN first (hipcc line at the top of tools/cabi_bench.cpp)"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for P in "$@"; do
  case $P in
    stats) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$BIN" $ARGS > "$OUT/stats.log" 2>&1
           find "$OUT/stats" -name '*kernel_stats.csv' -exec cat {} \; | cut -d, -f1-8 | head -12 ;;
    sq)    C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" ;;
    lds)   C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ;;
    mem)   C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" ;;
    ta)    C="TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUFFER_WAVEFRONTS_sum" ;;
    ta2)   C="TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" ;;
    *) echo "unknown pass $P"; continue ;;
  esac
  if [ "$P" != stats ]; then
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/$P" -- "$BIN" $ARGS > "$OUT/$P.log" 2>&1 || tail -3 "$OUT/$P.log"
  fi
done
python3 "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json" 2>/dev/null
python3 - "$OUT/summary.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception:
    sys.exit(0)
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-28s %.5g" % (c, x["mean_per_dispatch"]))
PY
