#!/bin/bash
# BASELINE config 4a (mixed lengths) with the row-packed kernels, instantiation by instantiation, on the GPU box: the launches
# serialised on one stream (HMK_NO_SIDE_STREAMS=1) under rocprofv3 -- kernel trace, then two PMC passes (own runs,
# --kernel-trace only beside them) -- and each instantiation's time against its LDS-cycle ideal (tools/rows_ideal.py).
#   gpurun -- 'bash tools/profile_config4a_rows.sh'   -> gpurun_out/round3/round3_config4a_lds_ideal.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round3
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp HMK_NO_SIDE_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/ser4a" -o s -- python3 "$R/tools/run_config4a.py" > "$O/ser4a.log" 2>&1 || exit 1
if [ -z "$NOPMC" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$O/pmc4a_i" -o s -- python3 "$R/tools/run_config4a.py" > "$O/pmc4a_i.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc4a_b" -o s -- python3 "$R/tools/run_config4a.py" > "$O/pmc4a_b.log" 2>&1 || exit 1
fi
cd "$R" && python tools/rows_ideal.py $(find "$O/ser4a" -name "*kernel_trace.csv" | head -1) $(find "$O"/pmc4a_* -name "*counter_collection.csv" 2>/dev/null) > "$O/round3_config4a_lds_ideal.jsonl"
cat "$O/round3_config4a_lds_ideal.jsonl"
