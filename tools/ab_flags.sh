#!/bin/bash
# A/B on the GPU box: rebuilds the library with different compile-time knobs (VARIANTS="flags;flags;...", an empty entry = the
# defaults; default list: the wave-priority settings of hmk_device.h) and times the plain pass of
# the BASELINE workload and of config 4a; leaves the LAST variant built (run `make -C hammock_amd/csrc -B` afterwards).
#   gpurun -- 'VARIANTS=";-DHMK_ROW_AHEAD=2" bash tools/ab_flags.sh'
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-greedy --no-configs"
IFS=";" read -ra VARS <<< "${VARIANTS:-;-DHMK_SETPRIO=0;-DHMK_SETPRIO_DRAIN=0;-DHMK_SETPRIO=1;-DHMK_SETPRIO=3;;-DHMK_SETPRIO=0}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_neighbors.hip hammock_amd/csrc/k_neighbors_rows.hip
  make -C hammock_amd/csrc -j8 CXXFLAGS="$F $v" > gpurun_out/ab/make.log 2>&1 || exit 1
  $B > gpurun_out/ab/b.json 2>/dev/null
  timeout -k 10 100 python tools/run_config4a.py > gpurun_out/ab/c4a.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab/b.json')); c=json.load(open('gpurun_out/ab/c4a.json'))
print('[$v]', round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), '4a', round(c['ms_median'],3), round(c['ms_min'],3))"
done
