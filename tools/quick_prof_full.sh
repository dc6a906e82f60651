cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_full
MTD_NO_SIDE_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-generator > $O/prof_full.log 2>&1
f=$(find $O/prof_full -name "*.db" | head -1); python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps 7
python - <<P
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$O/prof_full/*kernel_stats.csv")[0])))
tot=sum(float(r["MsPerStep"]) for r in rows)
for r in rows[:48]: print(r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:58].ljust(58), str(int(r["Calls"])//7).rjust(5), r["AverageNs"].rjust(10), r["MsPerStep"])
print("total", tot)
P
