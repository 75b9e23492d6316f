#!/bin/bash
# the plain pass of the BASELINE workload for several column-run lengths -> gpurun_out/cols_per_tile.log
mkdir -p gpurun_out
{
for c in 8192 16384 32768 65536 32768; do
  echo -n "HMK_COLS_PER_TILE=$c  "
  HMK_COLS_PER_TILE=$c python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-greedy --no-configs 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
done
} > gpurun_out/cols_per_tile.log 2>&1
cat gpurun_out/cols_per_tile.log
