#!/usr/bin/env python3
"""GPU idle-gap histogram of the last `steps` steps of a rocprofv3 --kernel-trace rocpd database."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ks = db.execute("select start, end, name from kernels order by start").fetchall()
t1 = max(k[1] for k in ks)
span = t1 - ks[0][0]
# steady-state window: the last `steps` steps, estimated from the adamw launches (2 per step)
ad = [k[0] for k in ks if "adamw_kernel" in k[2]]
cut = ad[-2 * steps - 1] if len(ad) > 2 * steps else ks[0][0]
rs = [k for k in ks if k[0] >= cut]
cur = rs[0][1]
idle = []
for s, e, n in rs[1:]:
    if s > cur:
        idle.append((s - cur, n))
    cur = max(cur, e)
win = (rs[-1][1] - rs[0][0])
print(f"window {win / 1e6:.1f} ms ({steps} steps: {win / steps / 1e6:.2f} ms/step), kernel-time sum {sum(e - s for s, e, _ in rs) / steps / 1e6:.2f} ms/step, idle {sum(g for g, _ in idle) / steps / 1e6:.2f} ms/step in {len(idle) / steps:.0f} gaps/step")
for lo, hi in ((0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e9)):
    sel = [g for g, _ in idle if lo <= g < hi]
    print(f"  gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(sel) / steps:.0f}/step, {sum(sel) / steps / 1e6:.2f} ms/step")
c = collections.Counter()
for g, n in idle:
    if g >= 100e3:
        c[n[:60]] += g / steps / 1e6
print("  >=100 us gaps by following kernel:", [(k, round(v, 2)) for k, v in c.most_common(6)])
