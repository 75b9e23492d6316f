#!/bin/bash
# config 4a for several column-run lengths -> gpurun_out/cols_4a.log
mkdir -p gpurun_out
{
echo -n "default  "; python tools/run_config4a.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['tiles'], round(d['ms_median'],3), round(d['ms_min'],3))"
for c in 2048 4096 8192 16384 32768; do
  echo -n "HMK_COLS_PER_TILE=$c  "
  HMK_COLS_PER_TILE=$c python tools/run_config4a.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['tiles'], round(d['ms_median'],3), round(d['ms_min'],3))"
done
} > gpurun_out/cols_4a.log 2>&1
cat gpurun_out/cols_4a.log
