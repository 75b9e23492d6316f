#!/bin/bash
# Everything profiles/round4_* is made of, on the GPU box (about 8 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_round4.sh'   -> gpurun_out/round4/ (+ gpurun_out/profiles_new/ from collect_profiles.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round4
mkdir -p "$O"
cd "$R"
ROUND=round4 bash tools/collect_profiles.sh > "$O/collect_profiles.log" 2>&1; echo collect_profiles $?
timeout -k 10 200 python tools/bench_uniform.py 100000 > "$O/round4_uniform_lengths.jsonl" 2> /dev/null; echo uniform $?
timeout -k 10 200 python tools/bench_uniform.py 100000 6,8,10,11,13,14,16,17,18,19 > "$O/round4_uniform_lengths_all.jsonl" 2> /dev/null; echo uniform_all $?
for p in "12 3 20 60 26 14" "7 2 12 60" "9 2 15 60"; do timeout -k 10 100 python tools/rows_probe_lx.py $p 2> /dev/null; done > "$O/round4_rows_probe_thresholds.jsonl"; echo probe $?
timeout -k 10 100 python tools/run_fasta_neighbors.py tests/golden/antibodies.fa.gz > "$O/round4_antibodies_neighbors.json" 2> /dev/null; echo antibodies $?
timeout -k 10 300 python tools/greedy_phases.py 100000 300000 1000000 > "$O/round4_greedy_phases.jsonl" 2> /dev/null; echo phases $?
timeout -k 10 300 python tools/greedy_phases.py --sorted 100000 1000000 > "$O/round4_greedy_phases_default_order.jsonl" 2> /dev/null; echo phases_sorted $?
timeout -k 10 300 python tools/greedy_phases_fasta.py tests/golden/antibodies.fa.gz > "$O/round4_greedy_phases_antibodies.jsonl" 2> /dev/null; echo phases_antibodies $?
timeout -k 10 120 python tools/run_config4a.py > "$O/round4_config4a.json" 2> /dev/null; echo 4a $?
timeout -k 10 200 python tests/tools/e2e_compare.py 100000 16 > "$O/round4_end_to_end_1e5.json" 2> /dev/null; echo e2e $?
timeout -k 10 300 python tools/px_step_time.py > "$O/round4_px_step_time.jsonl" 2> /dev/null; echo px $?
cd /tmp && export TMPDIR=/tmp
# the neighbour pass on the reference's antibodies example: kernel stats + LDS / VALU / HBM counters
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_ab" -o ab -- python3 "$R/tools/run_fasta_neighbors.py" "$R/tests/golden/antibodies.fa.gz" > /dev/null 2> "$O/prof_ab.log"; echo prof_ab $?
cp $(find "$O/prof_ab" -name "*kernel_stats.csv" | head -1) "$O/round4_antibodies_kernel_stats.csv"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"; do
    tag=$(echo "$c" | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_ab_$tag" -o pmc -- python3 "$R/tools/run_fasta_neighbors.py" "$R/tests/golden/antibodies.fa.gz" 4 > /dev/null 2> "$O/pmc_ab_$tag.log"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profmillion" -o m -- python3 "$R/tools/run_million.py" > "$O/round4_million.json" 2> "$O/profmillion.log"; echo profmillion $?
cp $(find "$O/profmillion" -name "*kernel_stats.csv" | head -1) "$O/round4_million_kernel_stats.csv"
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for path in sorted(glob.glob(os.path.join(out, "pmc_ab_*/**/*counter_collection.csv"), recursive=True)):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if "k_neighbors" in row["Kernel_Name"]:
                acc.setdefault((row["Kernel_Name"][:70], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
res = {}
for (k, c), v in acc.items():
    res.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in res.items():
    if "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        d["cycles_per_xcd"] = cyc
        if "SQ_LDS_IDX_ACTIVE" in d: d["lds_busy_frac"] = d["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        if "SQ_ACTIVE_INST_VALU" in d: d["valu_busy_frac"] = d["SQ_ACTIVE_INST_VALU"] / (cyc * 256)
    if "WRITE_SIZE" in d and "FETCH_SIZE" in d:
        d["hbm_traffic_bytes_per_launch"] = d["WRITE_SIZE"] * 1024 + 2 * d["FETCH_SIZE"] * 1024   # gfx950: FETCH_SIZE reports half of wide streaming reads
res["_note"] = "rocprofv3 --pmc passes (separate runs, --kernel-trace only beside them) of tools/run_fasta_neighbors.py tests/golden/antibodies.fa.gz; per-launch means"
json.dump(res, open(os.path.join(out, "round4_antibodies_pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
rm -rf "$O/prof_ab" "$O/profmillion" "$O"/pmc_ab_*
cp "$R"/gpurun_out/profiles_new/round4_* "$O/" 2>/dev/null
ls -la "$O"
