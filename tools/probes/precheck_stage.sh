#!/bin/bash
# precheck_ms of the 10^6 greedy call (generator order and the reference's default order) with and without the two-stage tables
mkdir -p gpurun_out
{
for lim in 100 300; do
  for flag in "" "--sorted"; do
    echo "== HMK_PRECHECK_TWO_STAGE_LIMIT=$lim $flag"
    HMK_PRECHECK_TWO_STAGE_LIMIT=$lim python tools/greedy_phases.py 1000000 $flag 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print({k: round(d[k], 2) for k in ('total_ms', 'score_ms', 'csr_ms', 'precheck_ms', 'device_loop_ms', 'phase1_ms')})
"
  done
done
} > gpurun_out/precheck_stage.log 2>&1
cat gpurun_out/precheck_stage.log
