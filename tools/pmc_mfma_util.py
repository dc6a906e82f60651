#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES
(tools/pmc_summary.py output).  SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader
engines, so util = MFMA_BUSY / (32 * SQ_BUSY); calibrated on tools/mfma_power (a register-only MFMA loop): 1.000.

  python tools/pmc_mfma_util.py summary.csv [out.csv]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mf = {r["Kernel_Name"]: r for r in rows if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"}
sq = {r["Kernel_Name"]: r for r in rows if r["Counter_Name"] == "SQ_BUSY_CYCLES"}
out = []
for k, m in mf.items():
    if k in sq and float(m["Total"]) > 0:
        out.append((k, int(m["Dispatches"]), float(m["Total"]), float(sq[k]["Total"]), float(m["Total"]) / (32.0 * float(sq[k]["Total"]))))
out.sort(key=lambda r: -r[2])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Kernel_Name", "Dispatches", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "MfmaUtil"])
for r in out:
    w.writerow([r[0][:140], r[1], round(r[2]), round(r[3]), round(r[4], 4)])
