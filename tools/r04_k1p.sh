#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export MULUT_NO_BUILD=1
timeout -k 10 300 python tools/prof_k1.py $2 > gpurun_out/${1}_k1_phases.txt 2> gpurun_out/${1}_k1_phases.err || { tail -3 gpurun_out/${1}_k1_phases.err; exit 1; }
cat gpurun_out/${1}_k1_phases.txt
