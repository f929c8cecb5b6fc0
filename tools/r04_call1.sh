#!/bin/bash
# round 4, GPU call 1: this round's starting point (bench line), the detailed path's counters on D-real, traces of small launches
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd "$R"
python -c 'from mulut_amd import _native; _native.build()' || exit 1
export MULUT_NO_BUILD=1
timeout -k 10 600 python bench.py > $O/r04a_bench.json 2> $O/r04a_bench.err; echo "bench rc=$?"
DIST=real bash tools/prof_round.sh r04a_real > $O/prof_r04a_real.log 2>&1; echo "prof real rc=$?"
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_c5 -- python3 $R/bench.py --config 5 --frames 8 --lr-h 270 --lr-w 480 --steps 50 > $O/small_c5.log 2>&1; echo "c5 small rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_set5 -- python3 $R/tools/small_call.py --h 128 --w 128 --reps 50 > $O/small_set5.log 2>&1; echo "set5 rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_270 -- python3 $R/tools/small_call.py --h 270 --w 480 --n 8 --reps 50 > $O/small_270.log 2>&1; echo "270 rc=$?" )
for d in small_c5 small_set5 small_270; do
  find $O/$d -name '*kernel_stats.csv' -exec cp {} $O/r04a_${d}_kernel_stats.csv \;
  f=$(find $O/$d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && tail -n 400 "$f" > $O/r04a_${d}_kernel_trace_tail.csv
  rm -rf $O/$d
done
echo done
