#!/usr/bin/env python3
"""For every GPU idle gap in a rocprofv3 (--kernel-trace --hip-runtime-trace) rocpd database, tell whether the
kernel that ends the gap was enqueued by the host long before (dependency / launch latency) or just then (host-bound).

  python tools/gap_attribution.py results.db [steps]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
views = [r[0] for r in db.execute("select name from sqlite_master where type='view'")]
kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
rc = [r[1] for r in db.execute("pragma table_info(regions)")] if "regions" in views else []
print("views:", views)
print("regions cols:", rc)
ks = db.execute("select start, end, name, stack_id, corr_id, queue_id from kernels order by start").fetchall()
api = {}
if rc:
    key = "stack_id" if "stack_id" in rc else "corr_id"
    for name, s, e, k in db.execute(f"select name, start, end, {key} from regions"):
        if "Launch" in name or "launch" in name:
            api[k] = (s, e, name)
    print("launch regions:", len(api), "key", key)
t_end = max(k[1] for k in ks)
span = t_end - ks[0][0]
cut = t_end - span * 0.6
rs = [k for k in ks if k[0] >= cut]
cur = rs[0][1]
host_bound = dep_bound = 0.0
hb = collections.Counter()
dbn = collections.Counter()
unknown = 0.0
for s, e, name, stack, corr, q in rs[1:]:
    if s > cur:
        gap = s - cur
        a = api.get(stack if (rc and "stack_id" in rc) else corr)
        if a is None:
            unknown += gap
        else:
            lead = s - a[1]           # kernel start minus end of the launch call
            if a[1] > cur - 5000:     # launched after (or within 5 us before) the GPU went idle => host was late
                host_bound += gap
                hb[name[:50]] += gap
            else:
                dep_bound += gap
                dbn[name[:50]] += gap
    cur = max(cur, e)
n = steps * 0.6
print(f"idle per step: host-late {host_bound / n / 1e6:.2f} ms, queued-but-waiting {dep_bound / n / 1e6:.2f} ms, unknown {unknown / n / 1e6:.2f} ms")
print("host-late, by first kernel after the gap:")
for k, v in hb.most_common(12):
    print(f"  {v / n / 1e6:6.2f} ms  {k}")
print("queued-but-waiting, by first kernel after the gap:")
for k, v in dbn.most_common(12):
    print(f"  {v / n / 1e6:6.2f} ms  {k}")
