#!/bin/bash
# The CLI at 10^6 synthetic 12-mers on the GPU box, twice: whole-process time and the stages' timeline (HMK_CLI_TIMING=1).
#   gpurun -- 'bash tools/cli_million.sh'   -> gpurun_out/cli_million.log
N=${1:-1000000}
mkdir -p gpurun_out
python tools/make_fasta.py $N /tmp/m_$N.fa || exit 1
{
for round in 1 2; do
  rm -rf /tmp/out_$round
  echo "== run $round"; ( time HMK_CLI_TIMING=1 hammock_amd/bin/hammock-hip greedy -i /tmp/m_$N.fa -d /tmp/out_$round ) 2>&1 | grep -E "hammock-hip\]|Clustering time|real"
done
for f in initial_clusters.tsv initial_clusters_sequences.tsv initial_clusters_sequences_original_order.tsv input_statistics.tsv; do
  cmp /tmp/out_1/$f /tmp/out_2/$f && echo "identical: $f" || echo "DIFFERENT: $f"
done
} > gpurun_out/cli_million.log 2>&1
cat gpurun_out/cli_million.log
