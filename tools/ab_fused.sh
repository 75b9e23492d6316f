#!/bin/bash
# A/B of the clustering call's scoring pass at 1e5 (default order): rebuilds the row-packed parts (minimal shapes) per variant
mkdir -p gpurun_out/ab
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_neighbors_rows.h
  make -C hammock_amd/csrc -j8 ROWSFLAGS="-DHMK_ROWS_MINIMAL $v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  echo "[$v]" | tee -a gpurun_out/ab/fused.txt
  for e in "X=1" "HMK_NO_BAND=1"; do
  env $e python tools/greedy_phases.py --sorted 100000 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$e', {k: round(v,3) if isinstance(v,float) else v for k,v in d.items() if k in ('wall_ms','score_ms','csr_ms','wait_rows_ms','phase1_ms','device_loop_ms','total_ms')})" | tee -a gpurun_out/ab/fused.txt
  done
done
