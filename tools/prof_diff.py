#!/usr/bin/env python3
"""Per-kernel difference of two rocpd_stats tables (ms per step): python tools/prof_diff.py before.csv after.csv [N]"""
import csv
import sys

a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ms = lambda t, k: float(t[k]["MsPerStep"]) if k in t else 0.0
print("total ms/step", round(sum(ms(a, k) for k in a), 3), "->", round(sum(ms(b, k) for k in b), 3))
for k in sorted(set(a) | set(b), key=lambda k: -abs(ms(a, k) - ms(b, k)))[:top]:
    d = lambda t: f"{ms(t, k):7.3f} ms ({t[k]['Calls']:>5s} x {float(t[k]['AverageNs']) / 1e3:7.1f} us)" if k in t else "      -"
    print(f"{k.replace('(anonymous namespace)::', '')[:64]:64s} {d(a)}  ->  {d(b)}")
