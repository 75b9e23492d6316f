#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4d
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; rc=$?; echo pytest $rc
tail -5 "$O/pytest.log"
if grep -q "Memory access fault" "$O/pytest.log"; then exit 1; fi
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/bench_uniform.py 100000 > "$O/uniform.jsonl" 2> "$O/uniform.err"; echo uniform $?
cat "$O/uniform.jsonl"
timeout -k 10 100 python tools/rows_probe_lx.py 12 3 20 14 60 | tee "$O/probe12.json"
timeout -k 10 100 python tools/rows_probe_lx.py 9 2 15 60 | tee "$O/probe9.json"
timeout -k 10 100 python tools/rows_probe_lx.py 7 2 12 60 | tee "$O/probe7.json"
timeout -k 10 100 python tools/run_config4a.py | tee "$O/c4a.json"
