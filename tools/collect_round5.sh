#!/bin/bash
# Everything profiles/round5_* is made of, on the GPU box, in three parts (each fits one gpurun call):
#   gpurun --timeout 1100 -- 'bash tools/collect_round5.sh bench'     kernel trace + stats of the default bench command, the PMC passes
#                                                                     (HBM bytes; LDS / VALU activity), config 4b's trace + counters,
#                                                                     then the bench line itself (reads the fresh counter summary)
#   gpurun --timeout 1100 -- 'bash tools/collect_round5.sh passes'    uniform lengths 6..20, thresholds, antibodies (pass, stats, counters),
#                                                                     config 4a, the 7-mers' counters with and without hits, the exchange step
#   gpurun --timeout 1100 -- 'bash tools/collect_round5.sh calls'     the clustering call: phases at 1e5 / 3e5 / 1e6 in both orders, antibodies,
#                                                                     mixed lengths, eight contexts on the one card (+ per-device timeline),
#                                                                     kernel stats of the 1e6 call (one context and eight), end to end at 1e5
# Outputs: gpurun_out/round5/round5_*; copy into profiles/.  Separate --pmc passes only ever run beside --kernel-trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round5
PART=${1:-bench}
mkdir -p "$O"
stats_of() { cp "$(find "$1" -name "*kernel_stats.csv" | head -1)" "$2"; }
if [ "$PART" = bench ]; then
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o bench -- \
        python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-greedy --no-configs > "$O/round5_bench_under_rocprof.json" 2> "$O/trace.log"; echo trace $?
    for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"; do
        tag=$(echo "$c" | cut -d' ' -f1)
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$tag" -o pmc -- \
            python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-greedy --no-configs > /dev/null 2> "$O/pmc_$tag.log"; echo pmc_$tag $?
    done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_local" -o local -- \
        python3 "$R/tools/run_neighbors_local.py" > "$O/round5_neighbors_local.json" 2> "$O/trace_local.log"; echo local $?
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$O/pmc_local" -o pmc -- \
        python3 "$R/tools/run_neighbors_local.py" > /dev/null 2> "$O/pmc_local.log"; echo pmc_local $?
    cd "$R"
    python3 tools/pmc_summary.py "$O" round5
    stats_of "$O/trace_local" "$O/round5_neighbors_local_kernel_stats.csv"
    cp "$O/round5_pmc_summary.json" profiles/round5_pmc_summary.json            # bench.py reads roofline.traffic from here
    [ -f "$O/round5_neighbors_local_pmc.json" ] && cp "$O/round5_neighbors_local_pmc.json" profiles/
    timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$O/round5_bench_n1.json"; echo bench $?
    rm -rf "$O/trace" "$O/trace_local" "$O"/pmc_*/
elif [ "$PART" = passes ]; then
    cd "$R"
    timeout -k 10 300 python tools/bench_uniform.py 100000 6,7,8,9,10,11,12,13,14,15,16,17,18,19,20 > "$O/round5_uniform_lengths.jsonl" 2> /dev/null; cp "$O/round5_uniform_lengths.jsonl" "$O/round5_uniform_lengths_all.jsonl"; echo uniform $?
    for p in "12 3 20 60 26 14" "7 2 12 60" "9 2 15 60" "6 2 11 60" "8 2 14 60"; do timeout -k 10 100 python tools/rows_probe_lx.py $p 2> /dev/null; done > "$O/round5_rows_probe_thresholds.jsonl"; echo probe $?
    timeout -k 10 100 python tools/run_fasta_neighbors.py tests/golden/antibodies.fa.gz > "$O/round5_antibodies_neighbors.json" 2> /dev/null; echo antibodies $?
    timeout -k 10 120 python tools/run_config4a.py > "$O/round5_config4a.json" 2> /dev/null; echo 4a $?
    timeout -k 10 300 python tools/configs_only.py > "$O/round5_configs.txt" 2> /dev/null; echo configs $?
    timeout -k 10 300 python tools/px_step_time.py > "$O/round5_px_step_time.jsonl" 2> /dev/null; echo px $?
    bash tools/pmc_lx.sh 7 2 12 7mers_default_threshold > /dev/null 2>&1; echo pmc7 $?
    bash tools/pmc_lx.sh 7 2 60 7mers_no_hits > /dev/null 2>&1; echo pmc7nh $?
    bash tools/pmc_lx.sh 6 2 11 6mers_default_threshold > /dev/null 2>&1; echo pmc6 $?
    for t in 7mers_default_threshold 7mers_no_hits 6mers_default_threshold; do cp "$R/gpurun_out/pmc_lx/$t.json" "$O/round5_pmc_$t.json"; done
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_ab" -o ab -- python3 "$R/tools/run_fasta_neighbors.py" "$R/tests/golden/antibodies.fa.gz" > /dev/null 2> "$O/prof_ab.log"; echo prof_ab $?
    stats_of "$O/prof_ab" "$O/round5_antibodies_kernel_stats.csv"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_4a" -o c4a -- python3 "$R/tools/run_config4a.py" > /dev/null 2> "$O/prof_4a.log"; echo prof_4a $?
    stats_of "$O/prof_4a" "$O/round5_config4a_kernel_stats.csv"
    rm -rf "$O/prof_ab" "$O/prof_4a"
elif [ "$PART" = calls ]; then
    cd "$R"
    timeout -k 10 300 python tools/greedy_phases.py 100000 300000 1000000 > "$O/round5_greedy_phases.jsonl" 2> /dev/null; echo phases $?
    timeout -k 10 300 python tools/greedy_phases.py --sorted 100000 1000000 > "$O/round5_greedy_phases_default_order.jsonl" 2> /dev/null; echo phases_sorted $?
    timeout -k 10 200 python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz > "$O/round5_greedy_phases_antibodies.jsonl" 2> /dev/null; echo phases_antibodies $?
    timeout -k 10 200 python tools/greedy_phases_mixed.py > "$O/round5_greedy_phases_mixed.jsonl" 2> /dev/null; echo phases_mixed $?
    HMK_GREEDY_TIMING=1 timeout -k 10 300 python tools/greedy_phases.py 1000000 --devices=0,0,0,0,0,0,0,0 > "$O/round5_greedy_phases_eight_contexts_one_gpu.jsonl" 2> "$O/round5_eight_contexts_one_gpu_timeline.txt"; echo eight $?
    HMK_GREEDY_TIMING=1 timeout -k 10 300 python tools/greedy_phases.py --sorted 1000000 --devices=0,0,0,0,0,0,0,0 > "$O/round5_greedy_phases_eight_contexts_one_gpu_default_order.jsonl" 2> "$O/round5_eight_contexts_one_gpu_default_order_timeline.txt"; echo eight_sorted $?
    HMK_GREEDY_TIMING=1 timeout -k 10 100 python tools/greedy_phases.py --sorted 100000 > /dev/null 2> "$O/round5_greedy_timeline_1e5.txt"; echo timeline1e5 $?
    HMK_GREEDY_TIMING=1 timeout -k 10 200 python tools/greedy_phases.py --sorted 1000000 > /dev/null 2> "$O/round5_greedy_timeline_1e6_default_order.txt"; echo timeline1e6 $?
    timeout -k 10 200 python tests/tools/e2e_compare.py 100000 16 > "$O/round5_end_to_end_1e5.json" 2> /dev/null; echo e2e $?
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profmillion" -o m -- python3 "$R/tools/run_million.py" > "$O/round5_million.json" 2> "$O/profmillion.log"; echo profmillion $?
    stats_of "$O/profmillion" "$O/round5_million_kernel_stats.csv"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof8" -o m8 -- python3 "$R/tools/greedy_phases.py" 1000000 --devices=0,0,0,0,0,0,0,0 > /dev/null 2> "$O/prof8.log"; echo prof8 $?
    stats_of "$O/prof8" "$O/round5_eight_contexts_one_gpu_kernel_stats.csv"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof1e5" -o g -- python3 "$R/tools/greedy_phases.py" --sorted 100000 > /dev/null 2> "$O/prof1e5.log"; echo prof1e5 $?
    stats_of "$O/prof1e5" "$O/round5_greedy_1e5_kernel_stats.csv"
    rm -rf "$O/profmillion" "$O/prof8" "$O/prof1e5"
fi
ls -la "$O"
