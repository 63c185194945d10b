#!/bin/bash
# Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into the tracked profiles/<tag>_* files.
set -e
TAG=${1:-r3}
cd "$(dirname "$0")/.."
P=gpurun_out/prof_$TAG
cp $P/stats_bench/vm/*_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $P/stats_msm/vm/*_kernel_stats.csv profiles/${TAG}_msm_kernel_stats.csv
cp $P/stats_ntt/vm/*_kernel_stats.csv profiles/${TAG}_ntt_kernel_stats.csv
mkdir -p /tmp/pmc_$TAG && rm -rf /tmp/pmc_$TAG/* && for d in msm_fetch msm_write ntt_fetch ntt_write; do mkdir -p /tmp/pmc_$TAG/$d && cp $P/pmc/$d/vm/*counter_collection.csv /tmp/pmc_$TAG/$d/; done
python tools/pmc_summary.py /tmp/pmc_$TAG profiles/${TAG}_pmc_summary.txt profiles/${TAG}_pmc_traffic.json
python tools/sq_summary.py profiles/${TAG}_sq_counters.txt \
  "msm 2^20=$(ls $P/sq_msm/vm/*counter_collection.csv):13631456=msm_bucket_kernel" \
  "msm 2^20=$(ls $P/sq_msm/vm/*counter_collection.csv):1048576=msm_chunk_kernel" \
  "coset ntt 2^22=$(ls $P/sq_ntt/vm/*counter_collection.csv):4194304=ntt_pass_kernel*3"
grep -E "msm 2|ntt 2|prover_wall" $P/log.txt | cut -c1-160 > profiles/${TAG}_run_log.txt
ls -la profiles/${TAG}_*
