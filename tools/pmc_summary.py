#!/usr/bin/env python3
"""Average a PMC counter per kernel name from rocprofv3 --output-format csv (*counter_collection.csv).

  python tools/pmc_summary.py DIR [out.csv]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for row in csv.DictReader(open(f)):
        key = (row.get("Kernel_Name", "?"), row.get("Counter_Name", "?"))
        acc[key][0] += float(row.get("Counter_Value", 0) or 0)
        acc[key][1] += 1
rows = sorted(((k[0], k[1], v[0] / max(v[1], 1), v[1], v[0]) for k, v in acc.items()), key=lambda r: -r[4])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Kernel_Name", "Counter_Name", "AvgPerDispatch", "Dispatches", "Total"])
for r in rows:
    w.writerow([r[0][:160], r[1], round(r[2], 1), r[3], round(r[4], 1)])
