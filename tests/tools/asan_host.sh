#!/bin/bash
# Sanitizer pass over the CPU-side code (GPU ASAN is not available on the pool):
#  1. host greedy merge (hmk_greedy.cpp) on random thresholded graphs, ASAN+UBSAN, then TSAN with phase 1's window pool on
#  2. the oracle's C restatement under ASAN+UBSAN through tests/test_oracle.py
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Ihammock_amd/csrc -Iinclude \
    tests/tools/asan_merge_harness.cpp hammock_amd/csrc/hmk_greedy.cpp -lpthread -o "$T/harness"
"$T/harness" | tail -3
# the same harness under ThreadSanitizer with the windowed phase 1 forced on (4 threads, windows of 16 rows)
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Ihammock_amd/csrc -Iinclude \
    tests/tools/asan_merge_harness.cpp hammock_amd/csrc/hmk_greedy.cpp -lpthread -o "$T/harness_tsan"
HMK_PHASE1_THREADS=4 HMK_PHASE1_WINDOW=16 "$T/harness_tsan" > "$T/tsan.log" 2>&1 || true
if grep -q "WARNING: ThreadSanitizer" "$T/tsan.log"; then grep -A12 "WARNING: ThreadSanitizer" "$T/tsan.log" | head -40; exit 1; fi
tail -1 "$T/tsan.log"
make -C oracle >/dev/null
cp oracle/_build/libhammock_oracle.so "$T/orig.so"
gcc -std=c11 -O1 -g -fsanitize=address,undefined -fopenmp -shared -fPIC oracle/hammock_oracle.c -o oracle/_build/libhammock_oracle.so
trap 'cp "$T/orig.so" oracle/_build/libhammock_oracle.so' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle.py -x -q
