#!/bin/bash
# Everything a round's profiles/ entries come from, in one GPU call:   bash tools/evidence_round.sh r03
# (build/ubench_stream and build/ubench_ifetch are built in the authoring container: tools/asm_stats.sh stage_tube2_kernelILi2 &&
#  python tools/ubench/gen_stream_ubench.py build/asm/kernel.s > build/ubench_stream.hip && hipcc ... ; the probe library by
#  python tools/prof_phases.py --build-only)
#   gpurun_out/<tag>_bench.json (default bench.py line), _config4.json, _config5.json, _finetune_stats.csv + ft_pmc_<tag>/ (fine-tune step),
#   prof_<tag>/ (tools/prof_round.sh: kernel stats + PMC of the bench workload), <tag>_tube2_phases.json, <tag>_ubench_stream.txt, <tag>_ubench_ifetch.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
O=$R/gpurun_out
cd "$R"
export MULUT_NO_BUILD=1
timeout -k 10 120 build/ubench_stream > $O/${TAG}_ubench_stream.txt 2>&1; echo "ubench_stream rc=$?"
timeout -k 10 120 build/ubench_ifetch > $O/${TAG}_ubench_ifetch.txt 2>&1; echo "ubench_ifetch rc=$?"
python tools/make_issue_json.py $O/${TAG}_ubench_stream.txt $TAG > /dev/null; echo "valu_issue rc=$?"
timeout -k 10 300 python tools/prof_phases.py > $O/${TAG}_tube2_phases.json 2>> $O/${TAG}_bench.err; echo "phases rc=$?"
bash tools/prof_round.sh $TAG > $O/prof_${TAG}.log 2>&1; echo "prof_round rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft_$TAG -- python $R/bench.py --config 4 --steps 10 > $O/ft_$TAG.log 2>&1 )
find $O/ft_$TAG -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_finetune_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && for k in sq lds; do
    if [ $k = sq ]; then C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; fi
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/ft_pmc_$TAG/$k -- python $R/bench.py --config 4 --steps 6 > $O/ft_pmc_$TAG.$k.log 2>&1 || echo "ft pmc $k failed"
  done )
python tools/summarize_pmc.py $O/ft_pmc_$TAG > $O/ft_pmc_$TAG/pmc_summary.json 2>/dev/null
python - "$O/ft_pmc_$TAG/pmc_summary.json" "$TAG" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from mulut_amd import _native
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no fine-tune counters:", e); sys.exit(0)
ks = {k: {c: v["mean_per_dispatch"] for c, v in d[k].items()} for k in d if "ft_stage" in k}
json.dump({"source_hash": _native.source_hash(), "tag": sys.argv[2], "workload": "bench.py --config 4", "kernels": ks}, open("profiles/finetune_counters.json", "w"), indent=1)
print("finetune_counters:", list(ks))
PY
timeout -k 10 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --config 4 > $O/${TAG}_config4.json 2>> $O/${TAG}_bench.err; echo "config4 rc=$?"
timeout -k 10 300 python bench.py --config 5 --frames 8 > $O/${TAG}_config5.json 2>> $O/${TAG}_bench.err; echo "config5 rc=$?"
timeout -k 10 400 python bench.py --batch rolled --skip-other --skip-strips --cpu-crop 0 > $O/${TAG}_bench_rolled_batch.json 2>> $O/${TAG}_bench.err; echo "bench (rolled batch) rc=$?"
[ -x build/ubench_lds_atomic ] && timeout -k 10 120 build/ubench_lds_atomic > $O/${TAG}_ubench_lds_atomic.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && for d in real noise; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${TAG}_$d -- python $R/bench.py --dist $d --cpu-crop 0 --steps 4 --warmup 2 --skip-other --skip-strips > $O/stats_${TAG}_$d.log 2>&1
    find $O/stats_${TAG}_$d -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats_$d.csv \;
  done )
timeout -k 10 600 python tools/fuzz_parity.py --cases 1000 --seed 303 > $O/${TAG}_fuzz_parity.jsonl 2>&1; echo "fuzz parity rc=$?"
timeout -k 10 900 python tools/fuzz_parity.py --cases 3000 --seed 404 >> $O/${TAG}_fuzz_parity.jsonl 2>&1; echo "fuzz parity (3000) rc=$?"
timeout -k 10 300 python tools/fuzz_finetune.py --cases 150 --seed 303 > $O/${TAG}_fuzz_finetune.jsonl 2>&1; echo "fuzz finetune rc=$?"
cp profiles/kernel_counters.json profiles/hbm_traffic.json profiles/valu_issue.json profiles/finetune_counters.json $O/ 2>/dev/null
echo done
