#!/bin/bash
# further seeds of the randomized sweeps (about 10 minutes): -> gpurun_out/round4/round4_fuzz_more.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round4
mkdir -p "$O"; cd "$R"
F="$O/round4_fuzz_more.jsonl"; : > "$F"
run() { echo "{\"tool\": \"$*\"}" >> "$F"; timeout -k 10 ${T:-400} python "$@" 2> "$O/fuzz_more.err" | tail -1 >> "$F"; echo "$1 $?"; }
T=300 run tests/tools/fuzz_greedy.py 500 12
T=300 run tests/tools/fuzz_greedy.py 500 13
T=300 run tests/tools/fuzz_greedy.py 500 14
T=300 run tests/tools/fuzz_neighbors.py 2000 6
T=300 run tests/tools/fuzz_clinkage.py 500 5
T=300 run tests/tools/fuzz_local.py 800 5
cat "$F"
