#!/bin/bash
# the clustering call at 1e5 in the reference's default order under a few switches: which part of the call is what
run() { echo "[$1]"; env $1 HMK_GREEDY_TIMING=1 python tools/greedy_phases.py --sorted 100000 2> gpurun_out/r4i/err_$2.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,3) if isinstance(v,float) else v for k,v in d.items() if k in ('wall_ms','score_ms','csr_ms','wait_rows_ms','phase1_ms','precheck_ms','device_loop_ms','host_precheck_ms','sequential_ms','total_ms','phase1_stop_index')})"; }
mkdir -p gpurun_out/r4i
run "X=1" base
run "HMK_NO_BAND=1" noband
run "HMK_PLACE_EDGES=0" count
run "HMK_PLACE_EDGES=0 HMK_NO_BAND=1" count_noband
run "HMK_PHASE1_THREADS=4" p1t4
grep "phase 1\|hmk greedy" gpurun_out/r4i/err_base.txt | tail -12
