#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4c
mkdir -p "$O"
cd "$R"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; rc=$?; echo bench $rc
if grep -q "Memory access fault" "$O/bench.err"; then tail -3 "$O/bench.err"; exit 1; fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; echo pytest $rc
tail -5 "$O/pytest.log"
if grep -q "Memory access fault" "$O/pytest.log"; then exit 1; fi
VARIANTS=";-DHMK_ROWS_AHEAD=1;-DHMK_ROWS_STAGE_EXACT=1024" bash tools/ab_rows4.sh
