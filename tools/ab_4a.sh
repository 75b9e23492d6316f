#!/bin/bash
# A/B of the mixed-length pass (BASELINE config 4a) on the GPU box: full rebuild of the row-packed parts per variant
mkdir -p gpurun_out/ab
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_neighbors_rows.h
  make -C hammock_amd/csrc -j16 ROWSFLAGS="$v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  echo "[$v]" | tee -a gpurun_out/ab/c4a.txt
  timeout -k 10 100 python tools/run_config4a.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_median'],3), round(d['ms_min'],3))" | tee -a gpurun_out/ab/c4a.txt
done
