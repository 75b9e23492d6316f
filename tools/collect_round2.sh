#!/bin/bash
# Everything profiles/round2_* is made of (beside tools/collect_profiles.sh), on the GPU box (about 4 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_round2.sh'   -> gpurun_out/round2/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round2
mkdir -p "$O"
cd "$R"
timeout -k 10 300 python tests/tools/bench_configs.py > "$O/round2_configs.jsonl" 2> "$O/configs.err"; echo configs $?
timeout -k 10 120 python tools/run_config4a.py > "$O/round2_config4a.json" 2> /dev/null; echo 4a $?
timeout -k 10 300 python tools/greedy_phases.py 100000 300000 1000000 > "$O/round2_greedy_phases.jsonl" 2> /dev/null; echo phases $?
timeout -k 10 300 python tools/greedy_phases.py --sorted 100000 1000000 > "$O/round2_greedy_phases_default_order.jsonl" 2> /dev/null; echo phases_sorted $?
timeout -k 10 200 python tests/tools/e2e_compare.py 100000 16 > "$O/round2_end_to_end_1e5.json" 2> /dev/null; echo e2e $?
timeout -k 10 200 python tests/tools/e2e_antibodies.py 16 > "$O/round2_end_to_end_antibodies.json" 2> /dev/null; echo antibodies $?
timeout -k 10 200 python tests/tools/e2e_mixed_compare.py > "$O/round2_end_to_end_mixed_1e5.json" 2> /dev/null; echo mixed $?
timeout -k 10 200 python tests/tools/e2e_clinkage.py > "$O/round2_end_to_end_clinkage.jsonl" 2> /dev/null; echo clinkage $?
timeout -k 10 200 python tools/run_neighbors_local.py > "$O/round2_neighbors_local.json" 2> /dev/null; echo local $?
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof4a" -o c4a -- python3 "$R/tools/run_config4a.py" > "$O/prof4a.log" 2>&1; echo prof4a $?
cp $(find "$O/prof4a" -name "*kernel_stats.csv" | head -1) "$O/round2_config4a_kernel_stats.csv"
cd "$R" && timeout -k 10 300 python tools/px_step_time.py > "$O/round2_px_step_time.jsonl" 2> /dev/null; echo px $?
bash "$R/tools/profile_config4a.sh" > "$O/profile_config4a.log" 2>&1; echo lds_ideal $?
ls -la "$O"
