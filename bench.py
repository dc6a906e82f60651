#!/usr/bin/env python3
"""bench.py -- MTD-GAN hot-path throughput on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full_step|generator|inference512|patch_sampler]

One process per GPU (launched by torch.distributed.run for N > 1, backend nccl == RCCL).  A "step" is
one pass of the hot path over one batch of synthetic LDCT-shaped 64x64 patches (32 per GPU, weak
scaling).  Prints ONE JSON line on rank 0.  `value` = patches/s over all ranks, inputs resident in HBM.
Extra objects: `roofline` for the dominant kernel -- the fp32-MFMA implicit-GEMM instantiation with the
largest share of the step's GPU time -- timed by the library's own launch profiler (HIP events recorded
on the launch stream directly around that kernel, include/mtdgan_hip.h mtd_prof_*) in a second pass of
the same K steps with every kernel in one stream, so the events do not perturb `value` and a launch's duration is
the kernel's own rather than its share of a chip it runs on beside other streams' kernels; and `cpu_baseline` (the CPU oracle, kind "port",
timed on rank 0 at N == 1 on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 32
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA peak (dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("MTD_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


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
        nb, iters = 8, 2
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
    # full training iteration: oracle.train_step (engine.py:33-55 restated) on a small batch
    nb, iters = 4, 2
    x, y = orc.synthetic_ldct(nb, seed=1234)
    state = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items()}
    state.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=8).items()})
    g = torch.Generator().manual_seed(3)
    masks = [(torch.rand(nb, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    opt = {}
    orders = [[1, 2], [0, 2], [0, 1]]
    orc.train_step(state, opt, x, y, masks, orders)
    t0 = time.perf_counter()
    for _ in range(iters):
        orc.train_step(state, opt, x, y, masks, orders)
    dt = time.perf_counter() - t0
    return {"value": round(nb * iters / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"oracle full G+D+PCGrad+AdamW step, {iters} x {nb} patches (1 warm-up), torch CPU {torch.__version__}, {cores} threads"}


def pmc_traffic(kernel, tag=""):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summaries (profiles/r1_pmc_*.csv: one pass
    with FETCH_SIZE, one with WRITE_SIZE, values in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    bench.py cannot collect PMC counters itself; returns None when the summaries are absent."""
    import csv
    tot = 0.0
    for fn, mult in ((f"r1_pmc_{tag}fetch_size.csv", 2.0), (f"r1_pmc_{tag}write_size.csv", 1.0)):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            return None
        hit = None
        for row in csv.DictReader(open(path)):
            if kernel in row["Kernel_Name"]:
                hit = float(row["AvgPerDispatch"]) * 1024.0 * mult
                break
        if hit is None:
            return None
        tot += hit
    return round(tot)


def pmc_mfma_util(kernel, tag=""):
    """MFMA-pipe utilisation of `kernel` from the committed PMC pass (profiles/r1_pmc_mfma_util.csv: SQ_VALU_MFMA_BUSY_CYCLES /
    (32 x SQ_BUSY_CYCLES), calibrated at 1.000 on a register-only MFMA loop, tools/pmc_mfma_util.py); None when absent."""
    import csv
    path = os.path.join(ROOT, "profiles", f"r1_pmc_{tag}mfma_util.csv")
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if kernel in row["Kernel_Name"]:
            return float(row["MfmaUtil"])
    return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("MTD_FORCE_DP", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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

    for _ in range(args.warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps
    value = per_gpu_units * world * args.steps / dt

    roofline = None
    if not args.no_roofline:
        from mtd_gan_amd import kernels as K
        cap = 8192 * max(1, args.steps)
        step = getattr(wl, "step_eager", wl.step)          # events cannot be recorded inside a graph replay
        # This pass runs every kernel in ONE stream: with the side streams and task chains of the timed region a launch's
        # event-to-event time includes the share of the chip it cedes to concurrent kernels, which says nothing about the
        # kernel.  (profiles/*_kernel_stats_single_stream.csv is the rocprofv3 trace of the same mode.)
        K.set_concurrency(False)
        step()
        K.prof_enable(cap)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        recs = K.prof_collect(cap)
        K.prof_enable(0)
        K.set_concurrency(True)
        # an empty event pair on the same stream: what the bracketing itself adds to every sample
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        ev_over_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        by = {}
        for r in recs:
            if r["kernel"].startswith("igemm"):
                d = by.setdefault(r["kernel"], {"ms": 0.0, "flops": 0.0, "n": 0, "shapes": {}})
                d["ms"] += r["ms"]
                d["flops"] += r["flops"]
                d["n"] += 1
                key = f"M{r['M']}_N{r['N']}_C{r['C']}_T{r['taps']}_S{r['splitk']}"
                sh = d["shapes"].setdefault(key, [0.0, 0.0, 0])
                sh[0] += r["ms"]; sh[1] += r["flops"]; sh[2] += 1
        if by:
            name, d = max(by.items(), key=lambda kv: kv[1]["ms"])
            raw_ms = d["ms"]
            for dd in by.values():                          # remove the event-pair overhead from every launch
                dd["ms"] = max(dd["ms"] - ev_over_ms * dd["n"], 1e-6)
                for sh in dd["shapes"].values():
                    sh[0] = max(sh[0] - ev_over_ms * sh[2], 1e-6)
            pmc_tag = {"full_step": "", "generator_fwd_bwd": "generator_"}.get(wl.name)
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            top = sorted(d["shapes"].items(), key=lambda kv: -kv[1][0])[:4]
            roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        # committed PMC passes: the default workload (profiles/r1_pmc_*.csv) and the generator workload
                        # (profiles/r1_pmc_generator_*.csv, tools/pmc_generator.sh)
                        "traffic": pmc_traffic(name, pmc_tag) if pmc_tag is not None else None,
                        "mfma_util_pmc": pmc_mfma_util(name, pmc_tag) if pmc_tag is not None else None,
                        "launches_timed": d["n"], "avg_launch_us": round(1e3 * d["ms"] / d["n"], 2),
                        "avg_launch_us_with_event_pair": round(1e3 * raw_ms / d["n"], 2), "event_pair_us": round(1e3 * ev_over_ms, 2),
                        "flops_per_launch": round(d["flops"] / d["n"]), "measured": "second pass of the same steps, all kernels in one stream",
                        "share_of_step_gpu_ms": round(d["ms"] / args.steps, 3),
                        "top_shapes": {k: {"us": round(1e3 * v[0] / v[2], 2), "tflops": round(v[1] / v[0] / 1e9, 2), "launches_per_step": v[2] // args.steps}
                                       for k, v in top},
                        "other_igemm": {k: {"tflops": round(v["flops"] / v["ms"] / 1e9, 2), "ms_per_step": round(v["ms"] / args.steps, 3)}
                                        for k, v in by.items() if k != name}}

    if roofline is None and not args.no_roofline and hasattr(wl, "roofline_bytes_per_step"):
        # byte-moving workloads: algorithmic bytes of a step over the step time (the kernels are launch-latency bound
        # at this size; the figure says how far from the HBM roof a whole batch is, not a single kernel)
        gbs = wl.roofline_bytes_per_step / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "bbox_kernel + sample_kernel (whole step)", "achieved": round(gbs, 2), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(gbs / 8000.0, 5), "traffic": None, "bytes_per_step": wl.roofline_bytes_per_step}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl)

    if rank == 0:
        line = {"metric": "GAN train-step imgs/sec (G+D fwd+bwd) @ 64x64 patch", "value": round(value, 2), "unit": "img/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": getattr(wl, "dtype", "f32"), "data": "synthetic",
                "config": wl.config(world), "roofline": roofline, "cpu_baseline": cpu}
        line.update(wl.extra())
        if "algorithmic_gflop_per_patch" in line:          # whole-step rate against the same peak, from the timed region
            tf = line["algorithmic_gflop_per_patch"] * per_gpu_units * world / (ms * 1e-3) / 1e3 / world
            line["step_tflops_per_gpu"] = round(tf, 2)
            line["step_frac_of_fp32_mfma_peak"] = round(tf / PEAK_F32_MFMA_TFLOPS, 4)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
