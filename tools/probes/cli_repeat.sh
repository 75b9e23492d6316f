#!/bin/bash
# the CLI at 10^6 several times in a row, every step timed  -> gpurun_out/cli_repeat.log
#   bash tools/probes/cli_repeat.sh [N] [runs] [ENV=value ...]
N=${1:-1000000}; REPS=${2:-5}; shift; shift
mkdir -p gpurun_out
python tools/make_fasta.py $N /tmp/m_$N.fa || exit 1
{
for r in $(seq $REPS); do
  rm -rf /tmp/out_r; echo "== run $r $*"
  ( time env HMK_CLI_TIMING=1 "$@" hammock_amd/bin/hammock-hip greedy -i /tmp/m_$N.fa -d /tmp/out_r ) 2>&1 | grep -E "hammock-hip\]|\[hmk\]|Clustering time|real"
done
} >> gpurun_out/cli_repeat.log 2>&1
