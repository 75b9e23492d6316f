#!/bin/bash
# Everything profiles/round3_* is made of (beside tools/collect_profiles.sh), on the GPU box (about 6 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_round3.sh'   -> gpurun_out/round3/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round3
mkdir -p "$O"
cd "$R"
timeout -k 10 300 python tests/tools/bench_configs.py > "$O/round3_configs.jsonl" 2> "$O/configs.err"; echo configs $?
timeout -k 10 120 python tools/run_config4a.py > "$O/round3_config4a.json" 2> /dev/null; echo 4a $?
timeout -k 10 300 python tools/greedy_phases.py 100000 300000 1000000 > "$O/round3_greedy_phases.jsonl" 2> /dev/null; echo phases $?
timeout -k 10 300 python tools/greedy_phases.py --sorted 100000 1000000 > "$O/round3_greedy_phases_default_order.jsonl" 2> /dev/null; echo phases_sorted $?
timeout -k 10 300 python tools/greedy_phases.py --devices=0,0 100000 1000000 > "$O/round3_greedy_phases_two_contexts_one_gpu.jsonl" 2> /dev/null; echo phases_multi $?
timeout -k 10 300 python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz > "$O/round3_greedy_phases_antibodies.jsonl" 2> /dev/null; echo phases_antibodies $?
timeout -k 10 200 python tests/tools/e2e_compare.py 100000 16 > "$O/round3_end_to_end_1e5.json" 2> /dev/null; echo e2e $?
timeout -k 10 200 python tests/tools/e2e_antibodies.py 16 > "$O/round3_end_to_end_antibodies.json" 2> /dev/null; echo antibodies $?
timeout -k 10 200 python tests/tools/e2e_mixed_compare.py > "$O/round3_end_to_end_mixed_1e5.json" 2> /dev/null; echo mixed $?
timeout -k 10 200 python tests/tools/e2e_clinkage.py > "$O/round3_end_to_end_clinkage.jsonl" 2> /dev/null; echo clinkage $?
timeout -k 10 200 python tools/rows_probe.py 20 60 14 26 > "$O/round3_rows_probe_thresholds.json" 2> /dev/null; echo probe $?
HMK_GREEDY_TIMING=1 timeout -k 10 200 python tools/greedy_phases.py 1000000 2>&1 >/dev/null | grep "phase 1" > "$O/round3_phase1_breakdown.txt"; echo phase1 $?
HMK_GREEDY_TIMING=1 timeout -k 10 200 python tools/greedy_phases.py 1000000 --sorted 2>&1 >/dev/null | grep "phase 1" > "$O/round3_phase1_breakdown_default_order.txt"; echo phase1_sorted $?
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof4a" -o c4a -- python3 "$R/tools/run_config4a.py" > "$O/prof4a.log" 2>&1; echo prof4a $?
cp $(find "$O/prof4a" -name "*kernel_stats.csv" | head -1) "$O/round3_config4a_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profmillion" -o m -- python3 "$R/tools/run_million.py" > "$O/round3_million.json" 2> "$O/profmillion.log"; echo profmillion $?
cp $(find "$O/profmillion" -name "*kernel_stats.csv" | head -1) "$O/round3_million_kernel_stats.csv"
cd "$R" && timeout -k 10 300 python tools/px_step_time.py > "$O/round3_px_step_time.jsonl" 2> /dev/null; echo px $?
bash "$R/tools/profile_config4a_rows.sh" > "$O/profile_config4a.log" 2>&1; echo lds_ideal $?
rm -rf "$O/prof4a" "$O/profmillion" "$O/ser4a" "$O"/pmc4a_*
ls -la "$O"
