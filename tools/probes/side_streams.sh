#!/bin/bash
# config 4a (mixed lengths) with 1..8 side streams -> gpurun_out/side_streams.log
mkdir -p gpurun_out
{
for s in 1 2 3 4 6 8 3; do
  echo -n "HMK_SIDE_STREAMS=$s  "
  HMK_SIDE_STREAMS=$s python tools/run_config4a.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['ms_median'],3), round(d['ms_min'],3))"
done
} > gpurun_out/side_streams.log 2>&1
cat gpurun_out/side_streams.log
