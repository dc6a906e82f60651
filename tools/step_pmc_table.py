#!/usr/bin/env python3
"""Per-kernel utilisation table from the three PMC passes of tools/step_pmc_vmem.sh (gpurun_out/steppmc_{ta,tcp,sq}.csv):
matrix pipe busy, address unit (TA) busy, where the waves' cycles go (parked at s_waitcnt / barrier, stalled at issue, issuing)
and the L1 -> L2 read rate, all as shares of the kernel's own busy cycles summed over its dispatches.

  python tools/step_pmc_table.py [dir] > profiles/rN_full_step_pipes_per_kernel.txt"""
import collections
import csv
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def load(name):
    out = collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(d, name))):
        out[r["Kernel_Name"]][r["Counter_Name"]] = (int(r["Dispatches"]), float(r["Total"]))
    return out


ta, tcp, sq = load("steppmc_ta.csv"), load("steppmc_tcp.csv"), load("steppmc_sq.csv")
rows = sorted(((v["SQ_BUSY_CYCLES"][1] / 32, k) for k, v in sq.items() if "SQ_BUSY_CYCLES" in v), reverse=True)
total = sum(b for b, _ in rows)
print("full training step, eager single stream, 3 iterations (tools/step_pmc_vmem.sh); shares of each kernel's busy cycles")
print(f"{'kernel':60s} {'Mcycles':>8s} {'share':>6s} {'disp':>5s} {'MFMA busy':>9s} {'TA busy':>8s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'L1<-L2 B/clk/CU':>16s}")
for busy, k in rows[:40]:
    v = sq[k]
    wc = v["SQ_WAVE_CYCLES"][1]
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{name:60s} {busy / 1e6:8.2f} {100 * busy / total:5.1f}% {v['SQ_BUSY_CYCLES'][0]:5d} {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'][1] / 1024 / busy:8.1f}% "
          f"{100 * ta.get(k, {}).get('TA_TA_BUSY_sum', (0, 0))[1] / 256 / busy:7.1f}% {100 * v['SQ_WAIT_ANY'][1] / wc:6.1f}% {100 * v['SQ_WAIT_INST_ANY'][1] / wc:7.1f}% "
          f"{100 * v['SQ_ACTIVE_INST_ANY'][1] / wc:7.1f}% {tcp.get(k, {}).get('TCP_TCC_READ_REQ_sum', (0, 0))[1] * 128 / 256 / busy:16.1f}")
