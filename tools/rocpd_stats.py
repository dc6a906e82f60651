#!/usr/bin/env python3
"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database.

  python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv] [--steps K]

Writes the same columns as rocprofv3's kernel_stats.csv; with --steps also a per-step column."""
import csv
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = None
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
        args = [a for a in args if a != str(steps)]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(name, [])
        d.append(e - s)
    tot = sum(sum(v) for v in agg.values())
    out = []
    for name, v in agg.items():
        n = len(v)
        mean = sum(v) / n
        var = sum((x - mean) ** 2 for x in v) / n
        out.append([name, n, sum(v), round(mean, 3), round(100.0 * sum(v) / tot, 4), min(v), max(v), round(var ** 0.5, 3)])
    out.sort(key=lambda r: -r[2])
    hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    if steps:
        hdr.append("MsPerStep")
        for r in out:
            r.append(round(r[2] / steps / 1e6, 4))
    w = csv.writer(open(args[1], "w", newline="") if len(args) > 1 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(hdr)
    w.writerows(out)
    print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches", file=sys.stderr)


if __name__ == "__main__":
    main()
