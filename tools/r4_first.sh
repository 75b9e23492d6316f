#!/bin/bash
# round 4, first look at the new build on the GPU: tests, uniform lengths, thresholds, the bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4a
mkdir -p "$O"
cd "$R"
timeout -k 10 200 python tools/bench_uniform.py 100000 > "$O/uniform.jsonl" 2> "$O/uniform.err"; echo uniform $?
timeout -k 10 200 python tools/rows_probe.py 20 60 14 26 > "$O/probe.json" 2> "$O/probe.err"; echo probe $?
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; echo bench $?
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo pytest $?
tail -5 "$O/pytest.log"
cat "$O/uniform.jsonl" "$O/probe.json"
