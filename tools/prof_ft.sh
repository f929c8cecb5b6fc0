#!/bin/bash
# Kernel stats + two PMC passes of the fine-tune step (bench.py --config 4):   bash tools/prof_ft.sh <tag>
# -> gpurun_out/<tag>_finetune_stats.csv, gpurun_out/ft_pmc_<tag>/pmc_summary.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-ft}; O=$R/gpurun_out; mkdir -p $O
export MULUT_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft_$TAG -- python $R/bench.py --config 4 --steps 10 > $O/ft_$TAG.log 2>&1 || { echo "stats failed"; exit 1; }
find $O/ft_$TAG -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_finetune_stats.csv \;
for k in sq lds; do
  if [ $k = sq ]; then C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; fi
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/ft_pmc_$TAG/$k -- python $R/bench.py --config 4 --steps 6 > $O/ft_pmc_$TAG.$k.log 2>&1 || { echo "ft pmc $k failed"; exit 1; }
done
python $R/tools/summarize_pmc.py $O/ft_pmc_$TAG > $O/ft_pmc_$TAG/pmc_summary.json
python - $O/ft_pmc_$TAG/pmc_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in d:
    if "ft_stage" in k:
        print(k, {c: round(v["mean_per_dispatch"] / 1e6, 2) for c, v in d[k].items()})
PY
head -6 $O/${TAG}_finetune_stats.csv | cut -c1-120
