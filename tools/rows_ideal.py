#!/usr/bin/env python3
"""Per-instantiation LDS-cycle ideal of the mixed-length pass (BASELINE config 4a) for the row-packed kernels against measured
kernel times.  A pair of a class (la >= lb, d = la - lb) adds C = lb (2X + d + 1) - X (X + 1) cells (ShiftedScorer.java:67-77);
the row-packed kernel reads exactly C ds_read_b64 per 8 pairs, 2 LDS cycles per wave-instruction of 64 lanes
(MI355X_MICROARCH.md, LDS table), on 256 CUs at 2.4 GHz.
Usage: python tools/rows_ideal.py [kernel_trace.csv [counter_collection.csv ...]]"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hammock_amd.synth import synth_peptides

X = 3
res, off = synth_peptides(1, 100000, 7, 20)
L = np.diff(off.astype(np.int64))
cnt = np.bincount(L, minlength=33)
ideal = defaultdict(float)
pairs = defaultdict(int)
for la in range(1, 33):
    for lb in range(1, la + 1):
        n = cnt[la] * cnt[lb] if la != lb else cnt[la] * (cnt[la] - 1) // 2
        if n == 0:
            continue
        d = la - lb
        cap = 12 if lb <= 12 else 16 if lb <= 16 else 20
        cells = lb * (2 * X + d + 1) - X * (X + 1)
        clk = n / 64.0 / 8.0 * cells * 2.0          # CU-cycles
        ideal[(d, cap)] += clk / (256 * 2.4e9) * 1e3
        pairs[(d, cap)] += int(n)
PAT = r"k_neighbors_rows<(\d+), (\d+), (\d+), (?:false|true|0|1), (\d+), (\d+)>"
measured = {}
if len(sys.argv) > 1:
    durs = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(PAT, r["Kernel_Name"])
        if m:
            durs[(int(m.group(2)), int(m.group(3)), int(m.group(4)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for (d, cap, g), v in durs.items():
        measured[(d, cap)] = (g, float(np.median(v)))
counters = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        m = re.search(PAT, r["Kernel_Name"])
        if m:
            counters[(int(m.group(2)), int(m.group(3)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot_i = tot_m = 0.0
for k in sorted(ideal):
    row = {"d": k[0], "cap": k[1], "pairs": pairs[k], "ideal_ms": round(ideal[k], 4)}
    if k in measured:
        row.update(groups=measured[k][0], measured_ms=round(measured[k][1], 4), frac=round(ideal[k] / measured[k][1], 3))
        tot_m += measured[k][1]
    c = {name: float(np.median(v)) for name, v in counters.get(k, {}).items()}
    if "GRBM_GUI_ACTIVE" in c:
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256
        row["lds_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 3)
        row["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / cu_cycles, 3)
    if "SQ_INSTS_LDS" in c:
        waves = pairs[k] / 64.0
        row["lds_insts_per_64_pairs"] = round(c["SQ_INSTS_LDS"] / waves, 2)
        row["table_reads_per_64_pairs"] = round(ideal[k] * 1e-3 * 256 * 2.4e9 / 2.0 / waves, 2)
        row["valu_insts_per_64_pairs"] = round(c["SQ_INSTS_VALU"] / waves, 2)
    tot_i += ideal[k]
    print(json.dumps(row))
print(json.dumps({"ideal_ms_total": round(tot_i, 4), "measured_ms_sum_serialised": round(tot_m, 4), "frac": round(tot_i / tot_m, 3) if tot_m else None}))
