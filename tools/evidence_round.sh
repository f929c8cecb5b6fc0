#!/bin/bash
# Everything a round's profiles/ entries come from, in one GPU call:   bash tools/evidence_round.sh r02
#   gpurun_out/<tag>_bench.json (default bench.py line), _config4.json, _config5.json, _finetune_stats.csv (rocprofv3 kernel
#   stats of the fine-tune step), prof_<tag>/ (tools/prof_round.sh: kernel stats + PMC of the bench workload)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
O=$R/gpurun_out
cd "$R"
python -c 'from mulut_amd import _native; _native.build()' || exit 1
timeout -k 10 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --config 4 > $O/${TAG}_config4.json 2>> $O/${TAG}_bench.err; echo "config4 rc=$?"
timeout -k 10 300 python bench.py --config 5 > $O/${TAG}_config5.json 2>> $O/${TAG}_bench.err; echo "config5 rc=$?"
bash tools/prof_round.sh $TAG > $O/prof_${TAG}.log 2>&1; echo "prof_round rc=$?"
( cd /tmp && export TMPDIR=/tmp MULUT_NO_BUILD=1 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft_$TAG -- python $R/bench.py --config 4 --steps 10 > $O/ft_$TAG.log 2>&1 )
find $O/ft_$TAG -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_finetune_stats.csv \;
echo done
