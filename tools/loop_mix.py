#!/usr/bin/env python3
"""Instruction mix of each kernel's hottest basic block (the one with the most MFMAs) from a hipcc -save-temps assembly file:
beside fp32 MFMAs every vector-ALU instruction is paid in full, so VALU per MFMA is what a loop's efficiency is made of.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -c X.hip -o /tmp/x.o -save-temps=obj ; python tools/loop_mix.py /tmp/X-hip-amdgcn-amd-amdhsa-gfx950.s"""
import collections
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\S+):\s*; @", s, re.M)
for name in names:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    blocks = re.split(r"\n(\.LBB\d+_\d+):", s[i:j])
    best = None
    for k in range(1, len(blocks), 2):
        n = blocks[k + 1].count("v_mfma")
        if best is None or n > best[0]:
            best = (n, blocks[k + 1])
    if not best or best[0] < 8:
        continue
    c = collections.Counter()
    ops = collections.Counter()
    mf = ""
    for l in best[1].split("\n"):
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
            mf = op
        elif op.startswith("v_"):
            c["valu"] += 1
            ops[op] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("buffer_", "global_")):
            c["vmem"] += 1
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    cyc = 64 if "32x32x2" in mf else 32
    print(f"{dem[:46]:46s} mfma {c['mfma']:3d} ({mf[7:]:14s}) valu {c['valu']:4d} = {c['valu'] / max(c['mfma'], 1):4.1f} per mfma, ~{100 * c['valu'] * 3.5 / (c['mfma'] * cyc):3.0f} % of its clocks | salu {c['salu']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} | "
          + " ".join(f"{k[2:]}:{v}" for k, v in ops.most_common(6)))
