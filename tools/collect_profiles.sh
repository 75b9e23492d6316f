#!/bin/bash
# Re-collects the profile set bench.py's roofline block refers to (run on the GPU box):
#   gpurun -- 'bash tools/collect_profiles.sh'
# 1. kernel trace + stats of the default bench command (average duration of the hot kernel must agree
#    with roofline.kernel_ms), 2. separate --pmc passes for HBM read / write bytes (never combined with
#    a trace domain other than --kernel-trace).  Outputs under gpurun_out/profiles_new/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_new
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- \
    python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-greedy > "$O/bench_under_rocprof.json" 2> "$O/trace.log"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -o pmc -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-greedy > /dev/null 2> "$O/pmc_$c.log"
done
python3 "$R/bench.py" --steps 30 > "$O/bench_n1.json"
find "$O" -name "*.csv" | head -20
