#!/bin/bash
# a third set of seeds on the final build (about 15 minutes): -> gpurun_out/round4/round4_fuzz_more2.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round4
mkdir -p "$O"; cd "$R"
F="$O/round4_fuzz_more2.jsonl"; : > "$F"
run() { echo "{\"tool\": \"$*\"}" >> "$F"; timeout -k 10 ${T:-400} python "$@" 2> "$O/fuzz_more2.err" | tail -1 >> "$F"; echo "$1 $?"; }
T=300 run tests/tools/fuzz_greedy.py 500 15
T=300 run tests/tools/fuzz_greedy.py 500 16
T=300 run tests/tools/fuzz_greedy.py 500 17
HMK_LOOP_CHAIN=1 T=300 run tests/tools/fuzz_greedy.py 500 18
T=300 run tests/tools/fuzz_neighbors.py 2000 7
T=300 run tests/tools/fuzz_neighbors.py 2000 8
T=300 run tests/tools/fuzz_clinkage.py 500 6
T=300 run tests/tools/fuzz_local.py 800 6
cat "$F"
