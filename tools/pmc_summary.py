#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/collect_round5.sh bench into the files profiles/ keeps:

    <round>_kernel_stats.csv          rocprofv3's own per-kernel stats (all dispatches)
    <round>_kernel_steady_state.json  the hot kernel's average over the TIMED dispatches only (warm-up dropped)
    <round>_pmc_summary.json          per-launch means of every counter for the hot kernel + derived fractions;
                                      hbm_traffic_bytes_per_launch = WRITE_SIZE + 2 x FETCH_SIZE (KB -> bytes): on gfx950
                                      FETCH_SIZE reports half of the bytes of wide streaming reads (MI355X_MICROARCH.md)
"""
import csv
import glob
import json
import os
import shutil
import sys

out, rnd = sys.argv[1], sys.argv[2]
HOT = os.environ.get("HMK_HOT_KERNEL_NAME", "k_neighbors_rows<3, 0, 12, true, 1, 0>")   # round 2: "k_neighbors_swar<2, 6, 2, 12, true, 0>"


def find(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return hits[0] if hits else None


stats = find("trace/**/*kernel_stats.csv")
if stats:
    shutil.copyfile(stats, os.path.join(out, f"{rnd}_kernel_stats.csv"))
trace = find("trace/**/*kernel_trace.csv")
if trace:
    durs = []
    with open(trace) as fh:
        for row in csv.DictReader(fh):
            if HOT in row["Kernel_Name"]:
                durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    durs.sort()
    timed = [d for _, d in durs[5:]]          # bench.py --warmup 5: the first five dispatches are warm-up
    with open(os.path.join(out, f"{rnd}_kernel_steady_state.json"), "w") as fh:
        json.dump({"kernel": HOT, "dispatches": len(durs), "warmup_dropped": 5, "timed": len(timed),
                   "steady_state_avg_ns": sum(timed) / max(len(timed), 1), "min_ns": min(timed) if timed else None,
                   "max_ns": max(timed) if timed else None,
                   "all_dispatches_avg_ns": sum(d for _, d in durs) / max(len(durs), 1)}, fh, indent=1)
# config 4b: k_neighbors_local_pk on this build (VALU instructions per DP cell, VALU busy fraction)
local_csv = find("pmc_local/**/*counter_collection.csv")
if local_csv:
    acc = {}
    with open(local_csv) as fh:
        for row in csv.DictReader(fh):
            if "k_neighbors_local" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    loc = {k: sum(v) / len(v) for k, v in acc.items()}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from hammock_amd.synth import synth_peptides
    _, off = synth_peptides(1, 100000, 7, 20)
    lens = np.diff(off.astype(np.int64)).astype(np.float64)
    cells = float(lens.sum()) ** 2 - float((lens * lens).sum())   # ordered pairs i != j
    if "SQ_INSTS_VALU" in loc:
        loc["valu_wave_instructions_per_64_cells"] = loc["SQ_INSTS_VALU"] / (cells / 64)
    if "GRBM_GUI_ACTIVE" in loc and "SQ_ACTIVE_INST_VALU" in loc:
        loc["valu_busy_frac"] = loc["SQ_ACTIVE_INST_VALU"] / (loc["GRBM_GUI_ACTIVE"] / 8 * 256)
    loc["dp_cells_per_launch"] = cells
    loc["_note"] = ("rocprofv3 --pmc pass of tools/run_neighbors_local.py (1e5 peptides of length 7..20, open -5, extend -1, thr 28); per-launch "
                    "means over the k_neighbors_local* dispatches; a lane carries two column sequences, so one wave instruction serves 128 cells")
    with open(os.path.join(out, f"{rnd}_neighbors_local_pmc.json"), "w") as fh:
        json.dump(loc, fh, indent=1)
    st = find("trace_local/**/*kernel_stats.csv")
    if st:
        shutil.copyfile(st, os.path.join(out, f"{rnd}_neighbors_local_kernel_stats.csv"))
summary = {}
for path in sorted(glob.glob(os.path.join(out, "pmc_[FWS]*/**/*counter_collection.csv"), recursive=True)):
    acc = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if HOT in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        summary[k] = sum(v) / len(v)
    tag = os.path.basename(os.path.dirname(os.path.dirname(path))) if "pmc_" in path else "pmc"
    shutil.copyfile(path, os.path.join(out, f"{rnd}_{[p for p in path.split(os.sep) if p.startswith('pmc_')][0]}.csv"))
if "WRITE_SIZE" in summary and "FETCH_SIZE" in summary:
    summary["hbm_traffic_bytes_per_launch"] = summary["WRITE_SIZE"] * 1024 + 2 * summary["FETCH_SIZE"] * 1024
    summary["hbm_write_bytes_per_launch"] = summary["WRITE_SIZE"] * 1024
    summary["hbm_read_bytes_per_launch_corrected"] = 2 * summary["FETCH_SIZE"] * 1024
pairs = 100000 * 99999 // 2
d = {}
if "GRBM_GUI_ACTIVE" in summary:
    cyc = summary["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
    d["cycles_per_xcd"] = cyc
    if "SQ_LDS_IDX_ACTIVE" in summary:
        d["lds_busy_frac"] = summary["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        d["lds_bank_conflict_per_active_cycle"] = summary.get("SQ_LDS_BANK_CONFLICT", 0) / summary["SQ_LDS_IDX_ACTIVE"]
    if "SQ_ACTIVE_INST_VALU" in summary:
        d["valu_busy_frac"] = summary["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 256 * 4)
if "SQ_INSTS_VALU" in summary:
    d["valu_insts_per_64_pairs"] = summary["SQ_INSTS_VALU"] / (pairs / 64)
if "SQ_INSTS_LDS" in summary:
    d["lds_insts_per_64_pairs"] = summary["SQ_INSTS_LDS"] / (pairs / 64)
summary["derived"] = d
summary["workload"] = "100000 synthetic length-12 peptides, BLOSUM62, X=3, p=0, thr=20"
summary["_note"] = ("rocprofv3 --pmc passes (separate runs, --kernel-trace only beside them) of `python3 bench.py --steps 3 --warmup 1 "
                    "--no-cpu-baseline --no-greedy`; per-launch means over the dispatches of " + HOT)
with open(os.path.join(out, f"{rnd}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps(summary, indent=1))
