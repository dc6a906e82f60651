#!/bin/bash
# Dynamic instruction counts per kernel of the full training step (one PMC pass, eager single stream): vector-ALU instructions
# beside MFMAs are paid in full on this chip, so (VALU - MFMA) per MFMA is each kernel's transform / address overhead.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MTD_GRAPH=0 MTD_NO_SIDE_STREAMS=1 MTD_LIST=0
O=gpurun_out; mkdir -p $O
rm -rf $O/steppmc_insts
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/steppmc_insts -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api > $O/steppmc_insts.log 2>&1 || { echo "pass failed"; tail -3 $O/steppmc_insts.log; exit 1; }
python tools/pmc_summary.py $O/steppmc_insts $O/steppmc_insts.csv
rm -rf $O/steppmc_insts
python - <<'PY'
import collections, csv
d = collections.defaultdict(dict)
for r in csv.DictReader(open("gpurun_out/steppmc_insts.csv")):
    d[r["Kernel_Name"]][r["Counter_Name"]] = (int(r["Dispatches"]), float(r["Total"]))
rows = sorted(((v["SQ_BUSY_CYCLES"][1] / 32, k) for k, v in d.items() if "SQ_BUSY_CYCLES" in v), reverse=True)
tot = sum(b for b, _ in rows)
print(f"{'kernel':58s} {'share':>6s} {'MFMA/wave':>10s} {'VALU/MFMA':>10s} {'ALU clocks : MFMA clocks':>26s} {'lanes busy':>11s}")
for busy, k in rows[:30]:
    v = d[k]
    mf, va = v["SQ_INSTS_VALU_MFMA_F32"][1], v["SQ_INSTS_VALU"][1] - v["SQ_INSTS_VALU_MFMA_F32"][1]
    waves = v["SQ_WAVES"][1]
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    # all the 32x32x2 kernels: 64 clocks per MFMA, ~3.5 per other vector instruction; lanes busy = both over the kernel's SIMD-cycles
    alu, mm = va * 3.5, mf * 64.0
    print(f"{name:58s} {100 * busy / tot:5.1f}% {mf / max(waves, 1):10.0f} {va / max(mf, 1):10.2f} {alu / max(mm, 1):26.2f} {100 * (alu + mm) / (busy * 1024):10.1f}%")
PY
