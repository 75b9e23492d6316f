#!/bin/bash
# The clustering call at 1e5 (default order) with the band hand-over's grid varied / left out (steady state = third call):
#   gpurun -- 'bash tools/band_probe.sh'
for e in "HMK_BAND_GRID=2" "HMK_BAND_GRID=4" "X=default" "HMK_BAND_GRID=32" "HMK_BAND_GRID=128" "HMK_BAND_GRID=512" "HMK_BAND_NO_HANDOVER=1" "HMK_NO_BAND=1" "X=default"; do
  env $e python tools/greedy_phases.py --sorted 100000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$e', {k: round(v,3) if isinstance(v,float) else v for k,v in d.items() if k in ('wall_ms','score_ms','csr_ms','wait_rows_ms','phase1_ms','device_loop_ms','total_ms')})"
done
