#!/bin/bash
# A/B of the CSR kernels at 10^6: rebuilds k_edges.hip with VARIANTS="flags;flags" and prints csr_ms of the steady-state calls
#   gpurun -- 'VARIANTS=";-DHMK_LB_LOADS=8;-DHMK_LB_CHUNK=131072" bash tools/probes/ab_csr.sh'
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_edges.hip
  make -C hammock_amd/csrc -j8 CXXFLAGS="$F $v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  python tools/greedy_phases.py ${N:-1000000} 2>/dev/null | python -c "
import sys, json
r = []
for l in sys.stdin:
    d = json.loads(l); p = d.get('phases_ms', d); r.append((round(p['csr_ms'], 2), round(p['total_ms'], 1)))
print('[$v]', r)" | tee -a gpurun_out/ab/csr_results.txt
done
