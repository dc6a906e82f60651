#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of a csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: tools/resource_usage.py conv_igemm [resfft4 ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mtd-gan_amd", "csrc")
KEYS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill"]


def table(name, extra=()):
    src = os.path.join(CSRC, name + ".hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null", *extra]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
            rows[cur] = {}
            continue
        for k in KEYS:
            m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
            if m and cur:
                rows[cur][k] = int(m.group(1))
    return rows


if __name__ == "__main__":
    for n in sys.argv[1:]:
        print("==", n)
        for k, v in table(n).items():
            print(f"{k[:90]:90s} V{v.get('VGPRs', 0):4d} A{v.get('AGPRs', 0):4d} scratch{v.get(KEYS[2], 0):5d} occ{v.get(KEYS[3], 0):2d} lds{v.get(KEYS[4], 0):7d}")
