#!/bin/bash
# A/B of the row-packed kernels on the GPU box: rebuilds the parts with different compile-time knobs (VARIANTS="flags;flags;...",
# an empty entry = the defaults; HMK_ROWS_MINIMAL shapes only: seconds per build) and times uniform 7-, 9- and 12-mers at the
# reference's defaults, a threshold nobody reaches, and 12-mers at threshold 14.  Leaves the LAST variant built: run
# `make -C hammock_amd/csrc -B` afterwards.  PROBES="L X thr thr ...;L X thr ..." overrides the probe list.
#   gpurun -- 'VARIANTS=";-DHMK_ROWS_STAGE=640" bash tools/ab_rows4.sh'
mkdir -p gpurun_out/ab
IFS=";" read -ra VARS <<< "${VARIANTS:-;}"
IFS=";" read -ra PRB <<< "${PROBES:-7 2 12 60;9 2 15;12 3 20 14 60}"
for v in "${VARS[@]}"; do
  touch hammock_amd/csrc/k_neighbors_rows.h
  make -C hammock_amd/csrc -j8 ROWSFLAGS="-DHMK_ROWS_MINIMAL $v" > gpurun_out/ab/make.log 2>&1 || { tail -5 gpurun_out/ab/make.log; exit 1; }
  echo "[$v]" | tee -a gpurun_out/ab/results4.txt
  for p in "${PRB[@]}"; do
    timeout -k 10 100 python tools/rows_probe_lx.py $p 2>/dev/null | tee -a gpurun_out/ab/results4.txt || exit 1
  done
done
