#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
TAG=${1:-r04q}
python -c 'from mulut_amd import _native; _native.build()' || exit 1
export MULUT_NO_BUILD=1
DIST=real bash tools/prof_round.sh ${TAG}_real > $O/prof_${TAG}_real.log 2>&1 || { echo "prof real failed"; exit 1; }
tail -1 $O/prof_${TAG}_real.log
cp profiles/kernel_counters_real.json profiles/hbm_traffic_real.json $O/ 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${TAG}_noise -- python $R/bench.py --dist noise --cpu-crop 0 --steps 4 --warmup 2 --skip-other --skip-strips > $O/stats_${TAG}_noise.log 2>&1
  find $O/stats_${TAG}_noise -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats_noise.csv \; ; rm -rf $O/stats_${TAG}_noise )
echo done
