#!/bin/bash
# The CLI at 10^6 synthetic 12-mers on the GPU box: whole-process time with the result files written side by side (default)
# and one after the other (HMK_CLI_SERIAL_WRITERS=1), and that both give the same bytes.
#   gpurun -- 'bash tools/cli_million.sh'   -> gpurun_out/cli_million.log
N=${1:-1000000}
mkdir -p gpurun_out
python tools/make_fasta.py $N /tmp/m_$N.fa || exit 1
{
for round in 1 2; do
  rm -rf /tmp/out_par /tmp/out_ser
  echo "== side by side"; ( time HMK_CLI_TIMING=1 hammock_amd/bin/hammock-hip greedy -i /tmp/m_$N.fa -d /tmp/out_par ) 2>&1 | grep -E "hammock-hip\]|Clustering time|real"
  echo "== serial";       ( time HMK_CLI_TIMING=1 HMK_CLI_SERIAL_WRITERS=1 hammock_amd/bin/hammock-hip greedy -i /tmp/m_$N.fa -d /tmp/out_ser ) 2>&1 | grep -E "hammock-hip\]|Clustering time|real"
done
for f in initial_clusters.tsv initial_clusters_sequences.tsv initial_clusters_sequences_original_order.tsv input_statistics.tsv; do
  cmp /tmp/out_par/$f /tmp/out_ser/$f && echo "identical: $f" || echo "DIFFERENT: $f"
done
} > gpurun_out/cli_million.log 2>&1
cat gpurun_out/cli_million.log
