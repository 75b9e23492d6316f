#!/usr/bin/env python3
"""Prints calls and average duration per kernel from a rocprofv3 *kernel_stats.csv (short names).  Usage: kernel_avgs.py file [substring ...]"""
import csv
import re
import sys

want = sys.argv[2:]
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", row["Name"]).replace("void ", "").replace("hmk::", "")
    if want and not any(w in name for w in want):
        continue
    print(f"{name:60s} calls {int(row['Calls']):4d}  avg {float(row['AverageNs']) / 1e6:9.3f} ms  total {float(row['TotalDurationNs']) / 1e6:9.2f} ms")
