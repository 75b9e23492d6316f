#!/bin/bash
# Re-collects the profile set bench.py's roofline block refers to (run on the GPU box):
#   gpurun -- 'bash tools/collect_profiles.sh'
# 1. kernel trace + stats of the default bench command (the hot kernel's average duration must agree with
#    roofline.kernel_ms; a steady-state-only average that drops the warm-up dispatches is written too),
# 2. separate --pmc passes (never combined with a trace domain other than --kernel-trace): HBM read / write bytes,
#    LDS / VALU activity,
# 3. the bench line itself, taken AFTER the counter summary exists (so roofline.traffic is this round's).
# Outputs under gpurun_out/profiles_new/ with the names profiles/ keeps (ROUND=round2).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-round3}
O=$R/gpurun_out/profiles_new
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- \
    python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-greedy --no-configs > "$O/${ROUND}_bench_under_rocprof.json" 2> "$O/trace.log"
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"; do
    tag=$(echo "$c" | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$tag" -o pmc -- \
        python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-greedy --no-configs > /dev/null 2> "$O/pmc_$tag.log"
done
# the LocalAlignmentScorer pass (config 4b): kernel stats + VALU counters of k_neighbors_local_pk on THIS build
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_local" -o local -- \
    python3 "$R/tools/run_neighbors_local.py" > "$O/${ROUND}_neighbors_local.json" 2> "$O/trace_local.log"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$O/pmc_local" -o pmc -- \
    python3 "$R/tools/run_neighbors_local.py" > /dev/null 2> "$O/pmc_local.log"
cd "$R"
python3 tools/pmc_summary.py "$O" "$ROUND"
cp "$O/${ROUND}_pmc_summary.json" profiles/${ROUND}_pmc_summary.json   # bench.py reads roofline.traffic from here
python3 bench.py --steps 20 --warmup 5 > "$O/${ROUND}_bench_n1.json"
ls -la "$O"
