#!/bin/bash
# round 4: tools/evidence_round.sh + what this round added (real-content counters, first-stage phase stamps, small-launch traces)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out
cd "$R"
python -c 'from mulut_amd import _native; _native.build()' || exit 1
bash tools/evidence_round.sh $TAG
export MULUT_NO_BUILD=1
DIST=real bash tools/prof_round.sh ${TAG}_real > $O/prof_${TAG}_real.log 2>&1; echo "prof real rc=$?"
cp profiles/kernel_counters_real.json profiles/hbm_traffic_real.json $O/ 2>/dev/null
timeout -k 10 300 python tools/prof_k1.py > $O/${TAG}_k1_phases.txt 2>> $O/${TAG}_bench.err; echo "k1 phases rc=$?"
[ -x build/ubench_stream_k1 ] && timeout -k 10 120 build/ubench_stream_k1 > $O/${TAG}_ubench_stream_k1.txt 2>&1
[ -x build/ubench_valu ] && timeout -k 10 200 build/ubench_valu > $O/${TAG}_ubench_valu_issue_cost.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_c5 -- python3 $R/bench.py --config 5 --frames 8 --lr-h 270 --lr-w 480 --steps 50 > $O/${TAG}_small_c5.log 2>&1; echo "c5 small rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_set5 -- python3 $R/tools/small_call.py --h 128 --w 128 --reps 50 > $O/${TAG}_small_set5.log 2>&1; echo "set5 rc=$?" )
for d in small_c5 small_set5; do
  find $O/$d -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_${d}_kernel_stats.csv \;
  f=$(find $O/$d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && tail -n 120 "$f" > $O/${TAG}_${d}_kernel_trace_tail.csv
  rm -rf $O/$d
done
echo all done
