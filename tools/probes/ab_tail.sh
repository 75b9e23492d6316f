#!/bin/bash
# A/B of the kernels after the scoring pass at 10^6 (k_edges.hip rebuilt with VARIANTS="flags;flags"): csr / pre-check / loop ms
#   gpurun -- 'VARIANTS=";-DHMK_PRE_UNROLL=8" bash tools/probes/ab_tail.sh'
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_edges.hip
  make -C hammock_amd/csrc -j8 CXXFLAGS="$F $v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  python tools/greedy_phases.py ${N:-1000000} $FLAGS 2>/dev/null | python -c "
import sys, json
r = []
for l in sys.stdin:
    d = json.loads(l); r.append(tuple(round(d[k], 2) for k in ('csr_ms', 'precheck_ms', 'device_loop_ms', 'total_ms')))
print('[$v]', r[1:])" | tee -a gpurun_out/ab/tail_results.txt
done
