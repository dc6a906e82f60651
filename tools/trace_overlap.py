#!/usr/bin/env python3
"""What runs BESIDE a kernel?  From a rocprofv3 kernel trace (rocpd SQLite database), for every dispatch whose name contains PATTERN in
the last iteration: its start / duration and the kernels whose execution intervals intersect it (name, overlap in us).

  python tools/trace_overlap.py x_results.db PATTERN [--all]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    rows = sorted(db.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    ends = marks[1::2]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    win = rows[lo:hi]
    t0 = win[0][1]
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    tot = ov = 0.0
    for n, s, e in win:
        if pat not in n:
            continue
        others = [(short(m), (min(e, e2) - max(s, s2)) / 1e3) for m, s2, e2 in win if (m, s2) != (n, s) and s2 < e and e2 > s]
        tot += (e - s) / 1e3
        ov += min((e - s) / 1e3, sum(o for _m, o in others))
        print(f"{(s - t0) / 1e3:10.1f} us +{(e - s) / 1e3:7.1f}  {short(n):40s} beside: " + ", ".join(f"{m} {o:.0f}" for m, o in others[:4]))
    print(f"total {tot:.1f} us, of which beside other kernels {ov:.1f} us")


if __name__ == "__main__":
    main()
