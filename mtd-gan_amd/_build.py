"""Builds libmtdgan_hip.so (gfx950) in-tree with hipcc.  No torch dependency in the library itself."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MTD_LAB_BUILD=1: the lab library (-DMTD_LAB: the kernel-selection environment switches of csrc/ are live), written beside
# the shipped one as libmtdgan_hip_lab.so and loaded only under MTD_LAB=1 (_options.py).  Never built by __graft_entry__.build().
LAB_BUILD = os.environ.get("MTD_LAB_BUILD", "0") == "1"
LIB = os.path.join(HERE, "libmtdgan_hip_lab.so" if LAB_BUILD else "libmtdgan_hip.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad.hip", "conv_c32_bwd.hip", "conv_winograd.hip", "conv_direct.hip", "resfft.hip", "resfft4.hip", "resfft_any.hip", "elementwise.hip",
           "specnorm.hip", "losses.hip", "metrics.hip", "sampler.hip", "pcgrad.hip", "adamw.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build_lab" if LAB_BUILD else "build")
    flags = FLAGS + (["-DMTD_LAB"] + os.environ.get("MTD_LAB_FLAGS", "").split() if LAB_BUILD else [])      # (MTD_LAB_FLAGS: e.g. -DW3_SKIP=1, lab probes)
    os.makedirs(objdir, exist_ok=True)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(HERE), "include", "mtdgan_hip.h"),
        os.path.join(CSRC, "conv_igemm.hip"), os.path.join(CSRC, "conv_wgrad.hip")]      # (conv_c32_bwd.hip / conv_winograd.hip include the two kernel files)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print("[mtdgan build]", " ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        elif verbose and out.strip():
            print(out.decode())
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[mtdgan build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
