#!/usr/bin/env python3
"""bench.py -- MTD-GAN hot-path throughput on MI355X (contract: see the task statement / DESIGN.md section 5).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full_step|generator|inference512|patch_sampler]

One process per GPU, backend nccl == RCCL.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
starts the N ranks itself (fresh child processes; the parent never touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.  A "step" is one pass of
the hot path over one batch of synthetic LDCT-shaped 64x64 patches (32 per GPU, weak scaling).  Prints ONE JSON line
on rank 0.  `value` = patches/s over all ranks, inputs resident in HBM.  Extra objects:
  roofline      the dominant kernel -- the fp32-MFMA implicit-GEMM instantiation with the largest share of the step's
                GPU time -- timed by the library's launch profiler (include/mtdgan_hip.h mtd_prof_*: two HIP events that
                ride on the kernel's own dispatch packet, i.e. the dispatch's begin / end timestamps, the duration a
                rocprofv3 kernel trace reports) in a second pass of the same K steps with every kernel in one stream;
  generator_fwd_bwd   BASELINE configs[1] (generator-only forward + backward, the north_star's 40 % target) timed in the
                same process after the main workload, with its own dominant-kernel figures;
  cpu_baseline  the CPU oracle (kind "port"), timed on rank 0 at N == 1 on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 32
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA peak (dense)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E
WINOGRAD_EXECUTED = 4.0 / 9.0     # csrc/conv_winograd.hip: 16 multiplications per 2 x 2 output tile instead of 36
WINOGRAD_F24_EXECUTED = 3.0 / 9.0     # ... and F(2x4, 3x3), wino_conv_kernel<.., 6>: 24 per 2 x 4 tile instead of 72


def executed_fraction(kernel):
    """Share of a layer's algorithmic multiplications (2 M N C 9) that the kernel's MFMAs execute."""
    if "wino" not in kernel:
        return 1.0
    if "wino32" in kernel or "wino_s2" in kernel:
        # F(3x3, 2x2) over the 4x4 / stride-2 layers: 16 of 36 multiplications per tile -- the ragged last tiles of a 32- / 16- / 8-pixel
        # row (6 / 27 / 27 % more) are NOT counted here, so these kernels' achieved figures are understated by that much
        return 16.0 / 36.0
    if "wino_conv3" in kernel:      # the split-bf16 kernel (opt-in): the same multiplications, each as six bf16 MFMA products -- counted
        return WINOGRAD_F24_EXECUTED if kernel.endswith("<6>") else WINOGRAD_EXECUTED      # here in fp32-equivalent flops
    return WINOGRAD_F24_EXECUTED if (kernel.endswith(", 6>") or "wino_c32" in kernel) else WINOGRAD_EXECUTED


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("MTD_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-generator", action="store_true", help="skip the configs[1] generator object of the default workload")
    ap.add_argument("--no-inference", action="store_true", help="skip the configs[4] whole-slice inference object of the default workload")
    ap.add_argument("--no-wino-split", action="store_true", help="skip the information-only leg that re-times the step on the split-bf16 Winograd kernel")
    ap.add_argument("--no-engine-api", action="store_true", help="skip the engine.train_MTD_GAN_Ours leg (profiling runs: only the timed steps)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect the roofline object's PMC counters in rocprofv3 child processes (the line then quotes the committed passes of profiles/)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only: gloo on the CPU, a trivial step (no GPU, no HIP library)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # launcher test: this rank exits with 3
    return ap.parse_args(argv)


# ====================================================================================================== launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Start --gpus fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment),
    wait for all of them and return the first non-zero exit code.  Runs before anything in this process has touched the
    GPU; nothing is exec'ed over a running process.  Rank 0's stdout (the JSON line) is this process's stdout."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = env0.get("MASTER_PORT") or str(_free_port())
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))

    def stop(signum, _frame):
        # the launcher was told to stop (scheduler, watchdog): ranks blocked in a collective would otherwise keep the GPUs
        for q in procs:
            if q.poll() is None:
                q.terminate()
        deadline = time.time() + 5.0
        while time.time() < deadline and any(q.poll() is None for q in procs):
            time.sleep(0.05)
        for q in procs:
            if q.poll() is None:
                q.kill()
        sys.exit(128 + signum)
    import signal
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop)
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:             # a rank died: the others would wait in a collective forever
                    q.terminate()
        if pending:
            time.sleep(0.05)
    return rc


# ====================================================================================================== CPU baseline
def cpu_baseline(wl):
    """The CPU oracle (oracle/mtdgan_oracle.py, kind "port": a restatement pinned bit-exact to the reference
    in the build container) timed on this box's host cores on a bounded sample of the same workload."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtdgan_oracle as orc
    # torch's CPU conv / FFT paths stop scaling (and on a 256-thread host collapse: 0.08 img/s measured with
    # every hardware thread) well before a full socket; 32 threads is the fastest setting found
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    if wl.name == "generator_fwd_bwd":
        nb, iters = 32, 2
        x, y = orc.synthetic_ldct(nb, seed=1234)
        gs = {k: v.clone().requires_grad_(True) for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items()}
        def run():
            out = orc.generator_forward(gs, x)
            torch.autograd.grad((out * y).sum(), list(gs.values()))
        run()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        dt = time.perf_counter() - t0
        return {"value": round(nb * iters / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
                "sample": f"oracle generator fwd+bwd, {iters} x {nb} patches (1 warm-up), torch CPU {torch.__version__}, {cores} threads"}
    if wl.name == "patch_sampler":
        import numpy as np
        import data_oracle as DO
        from mtd_gan_amd.create_datasets import Mayo
        lo, hi = wl.lo.cpu().numpy(), wl.hi.cpu().numpy()
        rng = np.random.RandomState(5)
        iters = 20
        t0 = time.perf_counter()
        for _ in range(iters):
            DO.window_patches(lo, hi, Mayo.draw_descriptors(rng, wl.n_slices))
        dt = time.perf_counter() - t0
        return {"value": round(32 * iters / dt, 1), "unit": "img/s", "cores": 1, "kind": "port",
                "sample": f"oracle/data_oracle.py (numpy restatement of the monai pipeline), {iters} batches of 32 patch pairs, 1 thread"}
    if wl.name == "inference512":
        g = orc.seeded_fill(orc.g_param_shapes(), seed=7)
        x, y = orc.synthetic_ldct(1, seed=1234, size=512)
        t0 = time.perf_counter()
        with torch.no_grad():
            out = orc.generator_forward(g, x)
            orc.psnr(out.clip(0, 1), y), orc.ssim(out.clip(0, 1), y), orc.rmse(out.clip(0, 1), y)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
                "sample": f"oracle generator forward + pixel metrics on one 512x512 slice, torch CPU {torch.__version__}, {cores} threads"}
    # full training iteration: oracle.train_step (engine.py:33-55 restated) on the workload's own batch of 32 patches:
    # one warm-up iteration, then two timed ones (about 10 s each on this box's host cores)
    nb, iters = PER_GPU_BATCH, 2
    x, y = orc.synthetic_ldct(nb, seed=1234)
    state = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items()}
    state.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=8).items()})
    g = torch.Generator().manual_seed(3)
    masks = [(torch.rand(nb, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    opt = {}
    orders = [[1, 2, 0], [0, 2, 1], [2, 0, 1]]
    orc.train_step(state, opt, x, y, masks, orders)
    per_iter = []
    for _ in range(iters):
        t0 = time.perf_counter()
        orc.train_step(state, opt, x, y, masks, orders)
        per_iter.append(time.perf_counter() - t0)
    dt = sum(per_iter)
    return {"value": round(nb * iters / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
            "seconds_per_iteration": [round(t, 2) for t in per_iter],
            "sample": f"oracle full G+D+PCGrad+AdamW step, 1 warm-up + {iters} timed x {nb} patches (the workload's batch), torch CPU {torch.__version__}, {cores} threads"}


# ====================================================================================================== PMC provenance
def _kernel_source_hash():
    """sha256 over the kernel sources: a committed PMC summary describes the kernels of ONE source state."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mtd-gan_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def _kernel_base(name, keep_template):
    """'void (anonymous namespace)::igemm_kernel<1, 1, 4, 1>((anonymous namespace)::IgemmParams)' -> 'igemm_kernel<1, 1, 4, 1>'
    (without the template arguments when the name looked for carries none)."""
    n = name.strip().replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth, end = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    n = n[:end].strip()
    return n if keep_template else n.split("<")[0]


def pmc_summary(kernel, tag):
    """HBM-side bytes per launch and MFMA-pipe utilisation of `kernel` from the committed rocprofv3 PMC summaries
    (separate passes: FETCH_SIZE, WRITE_SIZE in KiB with FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950;
    SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES), tools/pmc_mfma_util.py).  bench.py cannot collect PMC counters
    itself, so the figures carry their provenance (profiles/pmc_manifest.json: files, commit, kernel-source hash) and
    are emitted ONLY when the kernel sources are the ones the passes were taken on; otherwise null + the reason."""
    import csv
    mpath = os.path.join(ROOT, "profiles", "pmc_manifest.json")
    if not os.path.exists(mpath):
        return None, None, {"pmc": "no profiles/pmc_manifest.json"}
    man = json.load(open(mpath)).get(tag)
    if man is None:
        return None, None, {"pmc": f"no PMC passes recorded for workload '{tag}'"}
    prov = {"files": man["files"], "commit": man.get("commit"), "source_hash": man.get("source_hash")}
    if man.get("source_hash") != _kernel_source_hash():
        prov["stale"] = "kernel sources changed since the PMC passes were taken"
        return None, None, prov

    def lookup(fn, col):
        """Launch-weighted mean of `col` over EVERY row of the table whose kernel is `kernel` -- one row when the name carries its
        template arguments (one symbol), all instantiations of the family when it does not.  (Round 3 took the first matching
        row, and the first row differed from table to table.)  Returns (mean, dispatches) or (None, 0)."""
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            return None, 0
        tot, n = 0.0, 0
        for row in csv.DictReader(open(path)):
            if _kernel_base(row["Kernel_Name"], "<" in kernel) == kernel:          # exact name, not a substring hit
                d = int(float(row["Dispatches"]))
                tot += float(row[col]) * d
                n += d
        return (tot / n, n) if n else (None, 0)
    (f, nf), (w, nw) = lookup(man["files"]["fetch"], "AvgPerDispatch"), lookup(man["files"]["write"], "AvgPerDispatch")
    # MFMA utilisation of the family = busy cycles summed over its rows (tools/pmc_mfma_util.py: MFMA_BUSY / (32 x SQ_BUSY))
    mb = sb = 0.0
    nu = 0
    mpath2 = os.path.join(ROOT, "profiles", man["files"]["mfma"])
    for row in (csv.DictReader(open(mpath2)) if os.path.exists(mpath2) else ()):
        if _kernel_base(row["Kernel_Name"], "<" in kernel) == kernel:
            mb += float(row["SQ_VALU_MFMA_BUSY_CYCLES"])
            sb += float(row["SQ_BUSY_CYCLES"])
            nu += int(float(row["Dispatches"]))
    u = mb / (32.0 * sb) if sb else None
    prov["dispatches"] = {"fetch": nf, "write": nw, "mfma": nu}
    if u is not None:
        u = round(u, 4)
    traffic = round(2.0 * f * 1024.0 + w * 1024.0) if (f is not None and w is not None) else None
    return traffic, u, prov


def live_pmc_collect(workload):
    """PMC passes of this workload taken NOW, on this box, by this run -- BEFORE this process touches the GPU (so the children are
    ordinary child processes of a process without GPU state): three times `rocprofv3 --pmc <counter set> --kernel-trace --
    python3 bench.py --workload ... --steps 2` (FETCH_SIZE, WRITE_SIZE and the MFMA busy pair each in a pass of its own, eager
    single-stream launches so that every dispatch is attributed: what tools/measure.sh does).  Rank 0 at N = 1 only.  Any
    failure (no rocprofv3, a pass that fails or times out, this process itself running under a profiler) returns None and the
    line keeps the committed, hash-gated figures of profiles/.  Returns {"sums": {pass: {(kernel, counter): [total, dispatches]}},
    "seconds": s}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None                       # this run is being profiled itself: no nested profiler
    if os.environ.get("MTD_LAB", "0") == "1" and os.environ.get("MTD_LAB_LIB", "1") != "0":
        return None                       # this run may be timed on the lab library: children on the shipped one would measure another build
    # the children run the SHIPPED library with exactly the five switches below: every other MTD_* variable of the caller's shell
    # (a stray MTD_WINOGRAD=0 ...) is dropped, because MTD_LAB=1 would make it live in the children while the timed parent ignores it
    env = {k: v for k, v in os.environ.items() if not k.startswith("MTD_")}
    env.update(MTD_LAB="1", MTD_LAB_LIB="0", MTD_LIST="0", MTD_GRAPH="0", MTD_NO_SIDE_STREAMS="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got, t0 = {}, time.time()
    top = tempfile.mkdtemp(prefix="mtd_live_pmc_", dir="/tmp")
    try:
        for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"])):
            out = os.path.join(top, name)
            cmd = [exe, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1", "--no-roofline",
                   "--no-cpu-baseline", "--no-generator", "--no-inference", "--no-engine-api", "--no-wino-split", "--no-live-pmc"]
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, 9)      # the child's own process group (start_new_session): nothing else is in it
                except OSError:
                    pass
                proc.wait()
                return None
            if rc != 0:
                return None
            sums = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    c = sums.setdefault((row.get("Kernel_Name", ""), row.get("Counter_Name", "?")), [0.0, 0])
                    c[0] += float(row.get("Counter_Value", 0) or 0)
                    c[1] += 1
            if not sums:
                return None
            got[name] = sums
        return {"sums": got, "seconds": round(time.time() - t0, 1), "workload": workload}
    except Exception:
        return None
    finally:
        shutil.rmtree(top, ignore_errors=True)


def live_pmc_lookup(data, kernel):
    """(traffic bytes per launch, MFMA utilisation, provenance) of `kernel` from live_pmc_collect's passes, or None.  FETCH_SIZE /
    WRITE_SIZE are KiB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 -- the same rule as pmc_summary()."""
    if data is None:
        return None

    def fam(pass_name, counter):
        tot, n = 0.0, 0
        for (kn, cn), (t, d) in data["sums"].get(pass_name, {}).items():
            if cn == counter and _kernel_base(kn, "<" in kernel) == kernel:
                tot += t
                n += d
        return tot, n
    (f, nf), (w, nw) = fam("fetch", "FETCH_SIZE"), fam("write", "WRITE_SIZE")
    (mb, nm), (sb, _n) = fam("mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), fam("mfma", "SQ_BUSY_CYCLES")
    if not nf or not nw:
        return None
    traffic = round(2.0 * (f / nf) * 1024.0 + (w / nw) * 1024.0)
    util = round(mb / (32.0 * sb), 4) if sb else None
    prov = {"live": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES: one child process each, "
                    "--kernel-trace only) of `bench.py --workload %s --steps 2 --warmup 1`, eager single-stream launches, started by THIS run "
                    "on THIS box before its timed legs" % data["workload"],
            "source_hash": _kernel_source_hash(), "dispatches": {"fetch": nf, "write": nw, "mfma": nm}, "seconds": data["seconds"]}
    return traffic, util, prov


def apply_live_pmc(roof, data):
    """This run's own counters into a roofline object; the committed (builder-supplied, hash-gated) passes stay beside them."""
    if not roof or data is None:
        return
    lp = live_pmc_lookup(data, roof["kernel"])
    if lp is None:
        return
    roof["committed_pmc"] = {"traffic": roof.get("traffic"), "mfma_util_pmc": roof.get("mfma_util_pmc"), "pmc_provenance": roof.get("pmc_provenance")}
    roof["traffic"], roof["pmc_provenance"] = lp[0], lp[2]
    if lp[1] is not None:
        roof["mfma_util_pmc"] = lp[1]


# ====================================================================================================== roofline pass
def roofline_pass(wl, steps, pmc_tag):
    """Second pass of the same K steps, every kernel in one stream, the library's launch profiler on."""
    import torch
    from mtd_gan_amd import kernels as K
    cap = 8192 * max(1, steps)
    step = getattr(wl, "step_eager", wl.step)          # events cannot ride on launches inside a graph replay
    # One stream: beside the side streams and task chains of the timed region a kernel shares the chip with others, and
    # its duration then says nothing about the kernel.  (profiles/*_kernel_stats_single_stream.csv: rocprofv3, same mode.)
    K.set_concurrency(False)
    step()
    K.prof_enable(cap)
    attach = K.prof_mode(-1)
    K.FLOP_COUNT = {}
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    counted, K.FLOP_COUNT = K.FLOP_COUNT, None
    recs = K.prof_collect(cap)
    K.prof_enable(0)
    K.set_concurrency(True)
    by = {}
    for r in recs:
        if r["kernel"].startswith(("igemm", "c32_bwd", "wgrad", "wino")) or r["kernel"] in K.SPECTRAL_KERNELS:       # every profiled launch (the library's launch profiler)
            d = by.setdefault(r["kernel"], {"ms": 0.0, "flops": 0.0, "exec": 0.0, "bytes": 0.0, "n": 0, "shapes": {}})
            d["ms"] += r["ms"]
            d["flops"] += r["flops"]
            # the Winograd kernel executes 16 of the 36 multiplications per 2 x 2 output tile that the layer's definition counts
            d["exec"] += r["flops"] * executed_fraction(r["kernel"])      # (wino_conv_kernel, wgrad_wino_kernel)
            d["bytes"] += r["bytes"]
            d["n"] += 1
            key = f"M{r['M']}_N{r['N']}_C{r['C']}_T{r['taps']}_S{r['splitk']}"
            sh = d["shapes"].setdefault(key, [0.0, 0.0, 0])
            sh[0] += r["ms"]; sh[1] += r["flops"]; sh[2] += 1
    units = getattr(wl, "batch", None) or getattr(wl, "slices", PER_GPU_BATCH)
    executed = {k: round(v / steps / 1e9 / units, 4) for k, v in counted.items() if k not in ("launches", "conv_winograd_saved", "wgrad_winograd_saved")}
    executed_total = round(sum(executed.values()), 3)
    extra = {"executed_gflop_per_patch": executed_total, "executed_gflop_per_patch_by_kind": executed,
             "winograd_saved_gflop_per_patch": round((counted.get("conv_winograd_saved", 0.0) + counted.get("wgrad_winograd_saved", 0.0)) / steps / 1e9 / units, 4),
             "counted_launches_per_step": counted.get("launches", 0) // steps}
    if not by:
        return None, extra
    name, d = max(by.items(), key=lambda kv: kv[1]["ms"])
    if name in K.SPECTRAL_KERNELS:
        # HBM-bound kernel (the spectral path of whole-slice inference): algorithmic bytes of a launch -- every operand element
        # read once, every result written once -- over its measured duration, against the 8 TB/s of MI355X_MICROARCH.md
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        traffic, _util, prov = pmc_summary(name, pmc_tag) if pmc_tag is not None else (None, None, None)
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(d["bytes"] / d["n"]),
                    "pmc_provenance": prov, "launches_timed": d["n"], "avg_launch_us": round(1e3 * d["ms"] / d["n"], 2),
                    "timing": "HIP events on the kernel's dispatch packet" if attach == 1 else "HIP events recorded before / after the launch",
                    "measured": "second pass of the same steps, all kernels in one stream", "share_of_step_gpu_ms": round(d["ms"] / steps, 3),
                    "other_kernels": {k: ({"gb_per_s": round(v["bytes"] / v["ms"] / 1e6, 1)} if k in K.SPECTRAL_KERNELS else
                                          {"tflops": round(v["exec"] / v["ms"] / 1e9, 2)}) | {"ms_per_step": round(v["ms"] / steps, 3)}
                                      for k, v in by.items() if k != name}}
        return roofline, extra
    # `achieved` is what the matrix cores do per second -- executed MFMA flops, the quantity the fp32-MFMA peak bounds.  For the
    # implicit-GEMM kernels that is the layer's algorithmic count 2 M N C taps; the Winograd kernel multiplies 4 / 9 of it, and
    # its rate in algorithmic (direct-convolution-equivalent) flops -- which may exceed the peak -- is `algorithmic_tflops`.
    ach = d["exec"] / (d["ms"] * 1e-3) / 1e12
    top = sorted(d["shapes"].items(), key=lambda kv: -kv[1][0])[:4]
    traffic, util, prov = pmc_summary(name, pmc_tag) if pmc_tag is not None else (None, None, None)
    roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": round(d["bytes"] / d["n"]),
                "mfma_util_pmc": util, "pmc_provenance": prov,
                "launches_timed": d["n"], "avg_launch_us": round(1e3 * d["ms"] / d["n"], 2),
                "timing": ("HIP events on the kernel's dispatch packet (begin / end timestamps of the dispatch: no bracketing overhead, "
                           "nothing subtracted)" if attach == 1 else "HIP events recorded before / after the launch (includes the marker packets; nothing subtracted)"),
                "flops_per_launch": round(d["exec"] / d["n"]), "algorithmic_flops_per_launch": round(d["flops"] / d["n"]),
                "algorithmic_tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2),
                "flop_model": ("Winograd F(2x4,3x3): executed MFMA flops = 2 M N C 3 per launch (`achieved`, `flops_per_launch`, `frac`); the layer's "
                               "algorithmic count is 2 M N C 9 (`algorithmic_*`)" if executed_fraction(name) == WINOGRAD_F24_EXECUTED else
                               "Winograd F(2x2,3x3): executed MFMA flops = 2 M N C 4 per launch (`achieved`, `flops_per_launch`, `frac`); the layer's "
                               "algorithmic count is 2 M N C 9 (`algorithmic_*`)" if "wino" in name else "executed = algorithmic = 2 M N C taps"),
                "measured": "second pass of the same steps, all kernels in one stream",
                "share_of_step_gpu_ms": round(d["ms"] / steps, 3),
                "top_shapes": {k: {"us": round(1e3 * v[0] / v[2], 2), "tflops": round(v[1] / v[0] / 1e9 * executed_fraction(name), 2),
                                   "launches_per_step": v[2] // steps} for k, v in top},
                # (executed MFMA flops, like `achieved`: the two Winograd kernels at 4 / 9 of their layers' algorithmic count)
                "other_mfma_kernels": {k: ({"gb_per_s": round(v["bytes"] / v["ms"] / 1e6, 1)} if k in K.SPECTRAL_KERNELS else
                                           {"tflops": round(v["exec"] / v["ms"] / 1e9, 2)}) | {"ms_per_step": round(v["ms"] / steps, 3)}
                                       for k, v in by.items() if k != name}}
    # A launch whose algorithmic bytes need longer at the HBM peak than its executed flops at the MFMA peak is bounded by memory
    # (the 32 -> 32 channel layers of whole-slice inference: 805 MB against 12.9 GFLOP per launch): quote that roofline, keep the
    # matrix-pipe figures beside it.
    t_hbm, t_mfma = d["bytes"] / d["n"] / (PEAK_HBM_GBS * 1e9), d["exec"] / d["n"] / (PEAK_F32_MFMA_TFLOPS * 1e12)
    if t_hbm > t_mfma:
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         "bound_note": "algorithmic bytes / HBM peak = %.0f us > executed flops / fp32-MFMA peak = %.0f us per launch" % (t_hbm * 1e6, t_mfma * 1e6)})
    return roofline, extra


def timed(wl, steps, warmup, barrier):
    for _ in range(warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    barrier()
    return time.perf_counter() - t0


# ====================================================================================================== N > 1 diagnostics
def comm_diagnostics(wl, dev, world, steps, barrier, ms_step, avg_max):
    """What a reader of the N > 1 line needs beside `value` ("achieved all-reduce bus GB/s vs 7 x 153 GB/s"):
      allreduce_payload_mb   bytes one iteration all-reduces (three per-task shared-gradient vectors + the task-specific bucket of the
                             discriminator step + the generator's flat gradient)
      allreduce_alg_gbs /    a stand-alone all-reduce of four buffers of exactly those sizes (the three task vectors + the bucket),
      allreduce_bus_gbs      nothing else on the GPU, timed over `reps` rounds: bytes / time, and x 2 (N - 1) / N (RCCL's bus bandwidth:
                             what each rank's links carry; null at N = 1), against xGMI's 7 x 153 GB/s per GPU
      comm_exposed_ms        (timed step) - (the same step, same ranks, same launch list, with every data-path collective a no-op:
                             parallel.DataParallelSync.stub) -- the part of the collectives the backward kernels do NOT hide.
                             Measured LAST: the replicas drift apart while the collectives are stubbed
      launch_mode_per_rank   every rank's launch mode (a rank that fell back to eager launches would pace the others)
    Runs after the timed region; `avg_max(seconds)` -> max over ranks."""
    import torch
    import torch.distributed as dist
    dp = getattr(wl, "dp", None)
    if dp is None or not dist.is_initialized():
        return {}
    D, G = wl.model.Discriminator, wl.model.Generator
    n_shared = sum(p.numel() for p in D.shared_parameters())
    n_spec = sum(p.numel() for p in D.task_specific_parameters())
    n_gen = sum(p.numel() for p in G.parameters())
    out = {"allreduce_payload_mb": round(4 * (3 * n_shared + n_spec + n_gen) / 1e6, 2),
           "allreduce_messages_mb": [round(4 * n / 1e6, 2) for n in (n_shared, n_shared, n_shared, n_spec, n_gen)]}
    on_gpu = dist.get_backend() == "nccl"
    bufs = [torch.zeros(n, device=dev if on_gpu else None) for n in (n_shared, n_shared, n_shared, n_spec)]
    nbytes = 4 * (3 * n_shared + n_spec)

    def rounds(k):
        for _ in range(k):
            for b in bufs:
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
    rounds(2)
    reps = 5
    barrier()
    t0 = time.perf_counter()
    rounds(reps)
    barrier()
    t = avg_max(time.perf_counter() - t0) / reps
    alg = nbytes / t / 1e9
    out["allreduce_standalone_ms"] = round(1e3 * t, 3)
    out["allreduce_alg_gbs"] = round(alg, 1)
    out["allreduce_bus_gbs"] = round(alg * 2.0 * (world - 1) / world, 1) if world > 1 else None
    out["xgmi_peak_gbs_per_gpu"] = 7 * 153
    out["allreduce_backend"] = dist.get_backend() + (" (RCCL)" if on_gpu else " (through the host: rehearsal)")
    del bufs
    modes = [None] * world
    dist.all_gather_object(modes, wl.extra().get("launch_mode"))
    out["launch_mode_per_rank"] = modes
    # last: the same step with the collectives stubbed
    dp.stub = True
    try:
        sdt = avg_max(timed(wl, steps, 2, barrier))
    finally:
        dp.stub = False
    ms_stub = 1e3 * sdt / steps
    out["ms_per_step_collectives_stubbed"] = round(ms_stub, 3)
    out["comm_exposed_ms"] = round(ms_step - ms_stub, 3)
    return out


# ====================================================================================================== dry run
def dry_run(args, rank, world):
    """Launcher / rendezvous / max-over-ranks plumbing on the CPU (gloo): what tests/test_bench_launcher.py drives."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if rank == args.dry_run_fail_rank:
        return 3
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = dist.get_world_size() if world > 1 else 1
    v = torch.ones(1024)

    def step():
        w = v * 2.0
        if world > 1:
            dist.all_reduce(w)

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    def avg_max(sec):
        t = torch.tensor([sec], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    dt = avg_max(time.perf_counter() - t0)
    comm = {}
    if world > 1:
        # the N > 1 line's diagnostics through the same code as the GPU run (comm_diagnostics), on a stand-in workload whose
        # "model" has the partition methods it sizes the payload from and whose step all-reduces through a DataParallelSync
        from mtd_gan_amd import parallel

        class _Part(torch.nn.Module):
            def __init__(self, n_shared, n_spec):
                super().__init__()
                self.a, self.b = torch.nn.Parameter(torch.zeros(n_shared)), torch.nn.Parameter(torch.zeros(n_spec))

            def shared_parameters(self):
                return [self.a]

            def task_specific_parameters(self):
                return [self.b]

        class _Model:
            Discriminator, Generator = _Part(4096, 2048), _Part(512, 0)

        class _Wl:
            model, dp = _Model(), parallel.DataParallelSync(None)

            def step(self):
                w = v * 2.0
                self.dp.all_reduce_avg(w)

            def extra(self):
                return {"launch_mode": "dry run (rank %d)" % rank}
        comm = comm_diagnostics(_Wl(), None, world, args.steps, barrier, 1e3 * dt / args.steps, avg_max)
    if rank == 0:
        line = {"metric": "dry run (launcher plumbing, gloo on the CPU)", "value": round(PER_GPU_BATCH * seen * args.steps / dt, 2),
                "unit": "img/s", "n_gpus": seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "dry-run"}, "roofline": None, "cpu_baseline": None, "ranks_seen": seen}
        line.update(comm)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


# ====================================================================================================== main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)              # nothing below has run in this process: no GPU call, no torch import
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} in the environment but --gpus {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    # (before anything here touches the GPU) the default line's PMC passes, live: see live_pmc_collect
    live = {}
    if world == 1 and args.gpus == 1 and not args.no_roofline and not args.no_live_pmc and args.workload in ("auto", "full_step") \
            and os.environ.get("MTD_FORCE_DP", "0") != "1":
        live["full_step"] = live_pmc_collect("full_step")
        if live["full_step"] is not None:            # (a box where the first pass fails is not asked twice more)
            if not args.no_generator:
                live["generator"] = live_pmc_collect("generator")
            if not args.no_inference:
                live["inference512"] = live_pmc_collect("inference512")
    import torch
    import torch.distributed as dist
    # MTD_DP_SHARE_GPU=1: rehearsal of the N > 1 schedule on a one-GPU box -- every rank on device 0, gloo instead of RCCL
    # (RCCL refuses two ranks on one device).  The line then says so (config.parallelism) and is no scaling figure.
    share_gpu = os.environ.get("MTD_DP_SHARE_GPU", "0") == "1" and world > 1
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("MTD_FORCE_DP", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries exactly one JSON line
            # (the contract), so the file descriptor points at stderr while the communicator is created and first used
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                warm = torch.zeros(1, device=dev)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "mtd-gan_amd", "libmtdgan_hip.so")):
        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()
    from mtd_gan_amd import bench_workloads as BW
    wl = BW.make(args.workload, dev, rank, world, PER_GPU_BATCH)
    per_gpu_units = getattr(wl, "slices", PER_GPU_BATCH)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def avg_max(sec):
        if world == 1:
            return sec
        t = torch.tensor([sec], device=None if share_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = avg_max(timed(wl, args.steps, args.warmup, barrier))
    ms = 1e3 * dt / args.steps
    value = per_gpu_units * ranks_seen * args.steps / dt

    roofline, extra = None, {}
    pmc_tag = {"full_step": "full_step", "generator_fwd_bwd": "generator", "inference512": "inference512"}.get(wl.name)
    if not args.no_roofline and wl.name in ("full_step", "generator_fwd_bwd", "inference512"):
        roofline, extra = roofline_pass(wl, args.steps, pmc_tag)
    if roofline is None and not args.no_roofline and hasattr(wl, "roofline_bytes_per_step"):
        # byte-moving workloads: algorithmic bytes of a step over the step time (the kernels are launch-latency bound
        # at this size; the figure says how far from the HBM roof a whole batch is, not a single kernel)
        gbs = wl.roofline_bytes_per_step / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "bbox_kernel + sample_kernel (whole step)", "achieved": round(gbs, 2), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(gbs / 8000.0, 5), "traffic": None, "bytes_per_step": wl.roofline_bytes_per_step}

    # ---- BASELINE configs[1] beside the headline workload: generator-only forward + backward on the same 32 patches
    gen = None
    if wl.name == "full_step" and not args.no_generator:
        gw = BW.make("generator", dev, rank, world, PER_GPU_BATCH)
        gdt = avg_max(timed(gw, args.steps, args.warmup, barrier))
        gms = 1e3 * gdt / args.steps
        gtf = gw.gflop_per_patch * PER_GPU_BATCH / (gms * 1e-3) / 1e3
        gen = {"config": gw.config(ranks_seen), "value": round(PER_GPU_BATCH * ranks_seen * args.steps / gdt, 2), "unit": "img/s",
               "ms_per_step": round(gms, 3), "algorithmic_gflop_per_patch": gw.gflop_per_patch, "step_tflops_per_gpu": round(gtf, 2),
               "step_frac_of_fp32_mfma_peak": round(gtf / PEAK_F32_MFMA_TFLOPS, 4), "target_frac": 0.40}
        gen.update(gw.extra())
        if not args.no_roofline:
            groof, gextra = roofline_pass(gw, args.steps, "generator")
            gen["roofline"] = groof
            gen.update(gextra)
        del gw

    # ---- BASELINE configs[4] beside the headline workload: whole-slice generator inference, 8 x 512 x 512 (engine.py:89,129)
    inf = None
    if wl.name == "full_step" and not args.no_inference and world == 1:
        iw = BW.make("inference512", dev, rank, world, PER_GPU_BATCH)
        isteps, iwarm = min(args.steps, 20), min(args.warmup, 3)
        idt = timed(iw, isteps, iwarm, barrier)
        ims = 1e3 * idt / isteps
        itf = iw.gflop_per_patch * iw.slices / (ims * 1e-3) / 1e3
        inf = {"config": iw.config(ranks_seen), "value": round(iw.slices * isteps / idt, 2), "unit": "slice/s", "steps": isteps,
               "warmup": iwarm, "ms_per_step": round(ims, 3), "algorithmic_gflop_per_slice": iw.gflop_per_patch,
               "step_tflops_per_gpu": round(itf, 2), "step_frac_of_fp32_mfma_peak": round(itf / PEAK_F32_MFMA_TFLOPS, 4)}
        if not args.no_roofline:
            iroof, iextra = roofline_pass(iw, isteps, "inference512")
            inf["roofline"] = iroof
            inf.update(iextra)
        inf["pixel_metrics_last_batch"] = iw.last
        del iw

    # ---- information only (round 5): the SAME step with the 3x3 stride-1 trunk layers on the split-bf16 Winograd kernel
    # (mtd_set_option("wino_split", 1); csrc/conv_winograd_split.h: fp32-accurate products from six bf16 MFMAs each).  A second
    # model instance in this process, same box, after everything that feeds the headline has been measured; the option is put back.
    split_leg = None
    if wl.name == "full_step" and world == 1 and rank == 0 and not args.no_wino_split:
        from mtd_gan_amd import _lib as _L
        from mtd_gan_amd import kernels as _K

        def _set_split(on):
            _L.lib().mtd_set_option(b"wino_split", int(on))
            _K._wino_px_cache.clear()
            _K._igemm_ws_cache.clear()
            _K.weights_changed(None)
        _set_split(1)
        try:
            sw = BW.make("full_step", dev, rank, world, PER_GPU_BATCH)
            sdt = timed(sw, args.steps, args.warmup, barrier)
            split_leg = {"ms_per_step": round(1e3 * sdt / args.steps, 3), "value": round(PER_GPU_BATCH * args.steps / sdt, 2), "unit": "img/s",
                         "what": "the same full step with mtd_set_option('wino_split', 1): the Winograd products of the 3x3 stride-1 layers with "
                                 "N % 64 == 0 as six bf16 MFMA products of exact three-way bf16 splits, fp32 accumulation (error against "
                                 "float64 <= 2 x the fp32 MFMA kernel's: tests/test_kernels_gpu.py::test_winograd_conv_vs_torch[*bf16x3*]); "
                                 "NOT the headline configuration"}
            del sw
        finally:
            _set_split(0)

    # the same iterations through the kept API (engine.train_MTD_GAN_Ours, reference engine.py:26-76): the timed step above is that
    # loop's body, this is the loop itself.  (Before the CPU baseline: run after it -- 32 host threads that have just been busy --
    # this leg sporadically read 1 ms per iteration slower than the timed step, 31.2 against 30.2-30.4.)
    api_ms = None
    if rank == 0 and world == 1 and hasattr(wl, "engine_api_ms") and not args.no_engine_api:
        api_ms = wl.engine_api_ms(args.steps)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl)

    # ---- the N > 1 line's diagnostics (also under MTD_FORCE_DP=1 on one rank): after everything that feeds the line, since the
    # last of them runs the step with its collectives stubbed
    comm = {}
    if wl.name == "full_step" and dist.is_initialized():
        comm = comm_diagnostics(wl, dev, world, args.steps, barrier, ms, avg_max)

    if rank == 0 and wl.name == "full_step":
        apply_live_pmc(roofline, live.get("full_step"))
        apply_live_pmc(gen.get("roofline") if gen else None, live.get("generator"))
        apply_live_pmc(inf.get("roofline") if inf else None, live.get("inference512"))
    if rank == 0:
        line = {"metric": "GAN train-step imgs/sec (G+D fwd+bwd) @ 64x64 patch", "value": round(value, 2), "unit": "img/s",
                "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": getattr(wl, "dtype", "f32"), "data": "synthetic",
                "config": wl.config(ranks_seen), "roofline": roofline, "cpu_baseline": cpu}
        if share_gpu:
            line["config"]["parallelism"] += " (rehearsal: all ranks on ONE GPU, gloo through the host -- not a scaling figure)"
        line.update(wl.extra())
        line.update(extra)
        line["ranks_seen"] = ranks_seen
        line.update(comm)
        if api_ms is not None:
            ams = api_ms
            line["engine_api"] = {"ms_per_step": round(ams, 3), "value": round(per_gpu_units * 1e3 / ams, 2), "unit": "img/s",
                                  "what": "engine.train_MTD_GAN_Ours over the same resident batch, steps iterations, logging included"}
        if "algorithmic_gflop_per_patch" in line:          # whole-step rate against the same peak, from the timed region
            tf = line["algorithmic_gflop_per_patch"] * per_gpu_units / (ms * 1e-3) / 1e3
            line["step_tflops_per_gpu"] = round(tf, 2)
            line["step_frac_of_fp32_mfma_peak"] = round(tf / PEAK_F32_MFMA_TFLOPS, 4)
            if "executed_gflop_per_patch" in line:           # ... and the same for the flops the step actually executes
                tfe = line["executed_gflop_per_patch"] * per_gpu_units / (ms * 1e-3) / 1e3
                line["step_executed_tflops_per_gpu"] = round(tfe, 2)
                line["step_executed_frac_of_fp32_mfma_peak"] = round(tfe / PEAK_F32_MFMA_TFLOPS, 4)
        if gen is not None:
            line["generator_fwd_bwd"] = gen
        if inf is not None:
            line["inference512"] = inf
        if split_leg is not None:
            line["full_step_wino_split"] = split_leg
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
