#!/bin/bash
# Detailed-content path of the final stage, before / after: the full-table gather kernel (detail_kernel=1) against the
# anchor-slab path (default) on D-noise (8 x LR 1080x1920x3), kernel stats + SQ + LDS + texture-addresser / L1 counters.
#   usage (GPU box): bash tools/prof_detail.sh r02     -> gpurun_out/pq_<tag>_detail_{slab,gather}/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
A="--frames 8 --iters 3 --dist noise --luts $R/tests/golden/luts"
bash "$R/tools/prof_quick.sh" ${TAG}_detail_slab "$A" stats sq lds ta ta2 mem > "$R/gpurun_out/${TAG}_detail_slab.txt" 2>&1
bash "$R/tools/prof_quick.sh" ${TAG}_detail_gather "$A --tuning detail_kernel=1" stats sq ta ta2 mem > "$R/gpurun_out/${TAG}_detail_gather.txt" 2>&1
