#!/bin/bash
# csr_ms of the 10^6 greedy call for several partition grids  -> gpurun_out/csr_grid.log
mkdir -p gpurun_out
{
for g in 256 512 768 1024; do
  echo "== HMK_CSR_PARTITION_GRID=$g"
  HMK_CSR_PARTITION_GRID=$g python tools/greedy_phases.py 1000000 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); p = d.get('phases_ms', d)
    print({k: round(v, 2) for k, v in p.items() if k in ('total_ms', 'score_ms', 'csr_ms')})
"
done
} > gpurun_out/csr_grid.log 2>&1
cat gpurun_out/csr_grid.log
