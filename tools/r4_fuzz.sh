#!/bin/bash
# long randomized sweeps against the oracle on the final build (about 8 minutes): -> gpurun_out/round4/round4_fuzz.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round4
mkdir -p "$O"; cd "$R"
F="$O/round4_fuzz.jsonl"; : > "$F"
run() { echo "{\"tool\": \"$*\"}" >> "$F"; timeout -k 10 ${T:-400} python "$@" 2> "$O/fuzz.err" | tail -1 >> "$F"; echo "$1 $?"; }
T=500 run tests/tools/fuzz_neighbors.py 3000 4
T=300 run tests/tools/fuzz_neighbors.py 1500 5
T=400 run tests/tools/fuzz_greedy.py 600 4
T=300 run tests/tools/fuzz_clinkage.py 600 4
T=300 run tests/tools/fuzz_local.py 600 4
cat "$F"
