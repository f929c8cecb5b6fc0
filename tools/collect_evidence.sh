#!/bin/bash
# copies what tools/evidence_round.sh <tag> left under gpurun_out/ into profiles/ (run in the authoring container after the GPU call)
R=$(cd "$(dirname "$0")/.." && pwd); T=${1:-r03}; G=$R/gpurun_out; P=$R/profiles
cp $G/prof_$T/kernel_stats.csv $P/${T}_kernel_stats.csv; cp $G/prof_$T/pmc_summary.json $P/${T}_pmc_summary.json; cp $G/prof_$T/bench_under_stats.json $P/${T}_bench_under_stats.json
cp $G/${T}_bench.json $P/${T}_bench.json; cp $G/${T}_bench_rolled_batch.json $P/; cp $G/${T}_config4.json $P/${T}_config4_finetune_step.json; cp $G/${T}_config5.json $P/
cp $G/${T}_finetune_stats.csv $P/${T}_finetune_kernel_stats.csv; cp $G/ft_pmc_$T/pmc_summary.json $P/${T}_finetune_pmc_summary.json
cp $G/${T}_tube2_phases.json $G/${T}_ubench_stream.txt $G/${T}_ubench_ifetch.txt $G/${T}_ubench_lds_atomic.txt $G/${T}_kernel_stats_real.csv $G/${T}_kernel_stats_noise.csv $G/${T}_fuzz_parity.jsonl $G/${T}_fuzz_finetune.jsonl $P/
cp $G/kernel_counters.json $G/hbm_traffic.json $G/valu_issue.json $G/finetune_counters.json $P/
