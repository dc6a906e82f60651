#!/usr/bin/env python3
"""Where does an iteration's wall time go on the GPU?  From a rocprofv3 kernel trace (rocpd SQLite database):

  python tools/trace_gaps.py x_results.db [--steps K] [--top N]

Steps are delimited by the optimizer launches (two `adamw_kernel` dispatches per iteration); over the last K complete
iterations: wall time per iteration, the union of the kernels' busy intervals (what the chip was doing something for), idle time
(gaps with NO kernel running: launch latency, stream hand-offs), the sum of kernel durations (> busy where streams overlap),
dispatches per iteration, and the kernels ranked by time with their counts -- plus the launches shorter than 10 us."""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    args = [a for a in args if a not in (str(K), str(top))] or args
    db = sqlite3.connect(args[0])
    rows = sorted(db.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    if len(marks) < 2 * (K + 1):
        raise SystemExit(f"only {len(marks)} adamw_kernel dispatches in the trace")
    ends = marks[1::2]                                   # the generator's optimizer launch closes an iteration
    lo, hi = ends[-K - 1] + 1, ends[-1] + 1
    win = rows[lo:hi]
    wall = (win[-1][2] - rows[lo - 1][2]) / 1e6 / K
    busy, cur_s, cur_e = 0, None, None
    for _n, s, e in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for _n, s, e in win)
    print(f"{K} iterations: wall {wall:.3f} ms / iteration, busy (union) {busy / 1e6 / K:.3f}, idle {wall - busy / 1e6 / K:.3f}, "
          f"sum of kernel durations {total / 1e6 / K:.3f}, {len(win) / K:.0f} dispatches / iteration")
    agg = {}
    for n, s, e in win:
        d = agg.setdefault(n, [0, 0])
        d[0] += e - s
        d[1] += 1
    short = [(n, s, e) for n, s, e in win if e - s < 10000]
    print(f"dispatches shorter than 10 us: {len(short) / K:.0f} / iteration, {sum(e - s for _n, s, e in short) / 1e6 / K:.3f} ms / iteration")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        nm = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"  {t / 1e6 / K:8.3f} ms  {c / K:6.1f} x {t / c / 1e3:8.2f} us  {nm[:110]}")


if __name__ == "__main__":
    main()
