#!/bin/bash
# A/B on the GPU box: rebuild the library with different wave-priority settings and time the greedy call at 1e5 / 3e5
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p gpurun_out/ab
for v in "0 0" "2 1" "2 0" "0 0" "2 0"; do
  set -- $v
  touch hammock_amd/csrc/k_neighbors.hip
  make -C hammock_amd/csrc CXXFLAGS="$F -DHMK_SETPRIO=$1 -DHMK_SETPRIO_PLACE=$2" > gpurun_out/ab/make.log 2>&1 || exit 1
  timeout -k 10 120 python tools/greedy_phases.py 100000 100000 300000 > gpurun_out/ab/gp_$1_$2.jsonl 2>/dev/null
  python -c "
import json,sys
for l in open('gpurun_out/ab/gp_$1_$2.jsonl'):
    d=json.loads(l)
    if d['call']>0: print('prio $1 place $2', d['n'], d['call'], round(d['wall_ms'],2), round(d['score_ms'],2))"
done
