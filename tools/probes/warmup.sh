#!/bin/bash
# the bench line's ms_per_step for several warm-up counts (clock ramp of an idle GPU) -> gpurun_out/warmup.log
mkdir -p gpurun_out
{
for w in 3 5 10 20 40 3; do
  echo -n "--steps 20 --warmup $w  "
  python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-greedy --no-configs 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
done
} > gpurun_out/warmup.log 2>&1
cat gpurun_out/warmup.log
