#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_gen
MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o gen -- python3 bench.py --workload generator --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/prof_gen.log 2>&1
find $O/prof_gen -name "*.db" | while read f; do python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps 16; done
python - <<P
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$O/prof_gen/*kernel_stats.csv")[0])))
for r in rows[:12]: print(r["Name"][:64].ljust(64), int(r["Calls"])//16, r["AverageNs"], r["MsPerStep"])
print(sum(float(r["MsPerStep"]) for r in rows))
P
