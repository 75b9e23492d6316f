#!/usr/bin/env python3
"""Per-instantiation LDS-cycle ideal of the mixed-length pass (BASELINE config 4a) against measured kernel times.
A pair of a class (la >= lb) costs lb lookups of NW = ceil((2X + la - lb + 1) / 4) dwords = lb * NW / 2 ds_read_b64, each
2 LDS cycles per wave-instruction of 64 pairs (MI355X_MICROARCH.md, LDS table), on 256 CUs at 2.4 GHz.
Usage: python tools/planes_ideal.py [kernel_trace.csv [counter_collection.csv ...]]
(rocprofv3 --kernel-trace / --pmc passes of a serialised run, HMK_NO_SIDE_STREAMS=1: tools/profile_config4a.sh)"""
import csv
import json
import re
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
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
        nw = -(-(2 * X + la - lb + 1) // 4)
        lbmax = 12 if lb <= 12 else 16 if lb <= 16 else 20 if lb <= 20 else 32
        clk = n / 64.0 * lb * nw / 2.0 * 2.0          # CU-cycles
        ideal[(nw, lbmax)] += clk / (256 * 2.4e9) * 1e3
        pairs[(nw, lbmax)] += int(n)
measured = {}
if len(sys.argv) > 1:      # a rocprofv3 kernel trace: median duration per instantiation (the first passes run at a low clock)
    durs = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"k_neighbors_planes<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
        if m:
            durs[(int(m.group(1)), int(m.group(3)), int(m.group(2)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for (nw, lbmax, rows), v in durs.items():
        measured[(nw, lbmax)] = (rows, float(np.median(v)))
counters = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_neighbors_planes<(\d+), (\d+), (\d+)>", r["Kernel_Name"])
        if m:
            counters[(int(m.group(1)), int(m.group(3)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot_i = tot_m = 0.0
for k in sorted(ideal):
    row = {"nw": k[0], "lbmax": k[1], "pairs": pairs[k], "ideal_ms": round(ideal[k], 4)}
    if k in measured:
        row.update(rows_per_tile=measured[k][0], measured_ms=round(measured[k][1], 4), frac=round(ideal[k] / measured[k][1], 3))
        tot_m += measured[k][1]
    c = {name: float(np.median(v)) for name, v in counters.get(k, {}).items()}
    if "GRBM_GUI_ACTIVE" in c:      # busy fractions as in tools/pmc_summary.py: per CU-cycle (8 XCDs, 256 CUs)
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256
        row["lds_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 3)
        row["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / cu_cycles, 3)
        row["lds_conflict_per_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_INSTS_LDS" in c:
        waves = pairs[k] / 64.0
        row["lds_insts_per_64_pairs"] = round(c["SQ_INSTS_LDS"] / waves, 2)
        row["table_reads_per_64_pairs"] = round(ideal[k] * 1e-3 * 256 * 2.4e9 / 2.0 / waves, 2)
        row["valu_insts_per_64_pairs"] = round(c["SQ_INSTS_VALU"] / waves, 2)
    tot_i += ideal[k]
    print(json.dumps(row))
print(json.dumps({"ideal_ms_total": round(tot_i, 4), "measured_ms_sum_serialised": round(tot_m, 4), "frac": round(tot_i / tot_m, 3) if tot_m else None}))
