#!/bin/bash
# A/B of the row-packed kernel on the GPU box: rebuilds k_neighbors_rows.hip with different compile-time knobs
# (VARIANTS="flags;flags;...", an empty entry = the defaults) and times the plain pass of the BASELINE workload and, unless
# NO4A=1, of config 4a; leaves the LAST variant built (run `make -C hammock_amd/csrc -B` afterwards).
#   gpurun -- 'VARIANTS=";-DHMK_ROWS_G=1;-DHMK_ROWS_G=3" NO4A=1 bash tools/ab_rows.sh'
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab
B="python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-greedy --no-configs"
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_neighbors_rows.hip
  make -C hammock_amd/csrc -j8 CXXFLAGS="$F $v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  $B > gpurun_out/ab/b.json 2>/dev/null
  if [ -z "$NO4A" ]; then timeout -k 10 100 python tools/run_config4a.py > gpurun_out/ab/c4a.json 2>/dev/null; else echo '{"ms_median": 0, "ms_min": 0}' > gpurun_out/ab/c4a.json; fi
  python -c "
import json
d=json.load(open('gpurun_out/ab/b.json')); c=json.load(open('gpurun_out/ab/c4a.json'))
print('[$v]', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), '4a', round(c['ms_median'],3), round(c['ms_min'],3))" | tee -a gpurun_out/ab/results.txt
done
