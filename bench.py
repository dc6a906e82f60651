#!/usr/bin/env python3
"""bench.py -- MTD-GAN hot-path throughput on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full_step|generator]

One process per GPU (launched by torch.distributed.run for N > 1, backend nccl == RCCL).  A "step" is
one pass of the hot path over one batch of synthetic LDCT-shaped 64x64 patches (32 per GPU, weak
scaling).  Prints ONE JSON line on rank 0.  `value` = patches/s over all ranks, inputs resident in HBM.
Extra objects: `roofline` for the dominant kernel (fp32-MFMA implicit-GEMM 3x3 conv, 32->32 ch,
M = 32*64*64 pixels: 2.416 GFLOP per launch) timed with HIP events on the launch stream in a second,
event-instrumented pass of the same K steps (so the events do not perturb `value`), and `cpu_baseline`
(the CPU oracle, kind "port", timed on rank 0 at N == 1 on a bounded sample).
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
CONV_FLOPS = 2.0 * PER_GPU_BATCH * 64 * 64 * 32 * 288   # one 3x3 32->32 conv launch at B=32


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
    return wl.cpu_baseline_run(orc, cores)


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "mtd-gan_amd", "libmtdgan_hip.so")):
        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()
    from mtd_gan_amd import bench_workloads as BW
    wl = BW.make(args.workload, dev, rank, world, PER_GPU_BATCH)

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
    value = PER_GPU_BATCH * world * args.steps / dt

    roofline = None
    if not args.no_roofline:
        from mtd_gan_amd import kernels as K
        K.PROFILE = []
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        evs = [(tag, a.elapsed_time(b)) for tag, a, b in K.PROFILE if tag == "igemm_3x3_c32_n32"]
        K.PROFILE = None
        if evs:
            avg_ms = sum(e for _, e in evs) / len(evs)
            ach = CONV_FLOPS / (avg_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "igemm_kernel<2,1,4,1> (3x3 conv 32->32, M=131072)", "achieved": round(ach, 2),
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": None, "launches_timed": len(evs), "avg_launch_us": round(avg_ms * 1e3, 2),
                        "flops_per_launch": CONV_FLOPS}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl)

    if rank == 0:
        line = {"metric": "GAN train-step imgs/sec (G+D fwd+bwd) @ 64x64 patch", "value": round(value, 2), "unit": "img/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": wl.config(world), "roofline": roofline, "cpu_baseline": cpu}
        line.update(wl.extra())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
