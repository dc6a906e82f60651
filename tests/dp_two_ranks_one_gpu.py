#!/usr/bin/env python3
"""The product's data-parallel schedule at world size 2 on ONE GPU (reference train.py:93-98 is what it replaces).

    python tests/dp_two_ranks_one_gpu.py            # parent: starts two fresh rank processes, never touches the GPU itself

A gpurun box has one MI355X and RCCL does not put two ranks on one device, so the two ranks share device 0 and talk over
gloo (which moves device tensors through the host; parallel.DataParallelSync then sums and scales instead of ReduceOp.AVG).
Everything else is the schedule an 8-GPU run executes: train_step.DStepTape.run_pcgrad's three task passes one after the other,
each task vector's all-reduce enqueued on the collective side stream under the next pass (the last pass ships its finished slices while it runs), rank 0's projection order broadcast to every rank, the 158 MB
task-specific bucket under the Gram / combine kernels, the generator bucket.

What is checked, per rank: ONE iteration of engine.train_MTD_GAN_Ours on this rank's 16 patches of the 32-patch batch of
tests/golden/step_seeded_b32.json (the step the REFERENCE ran in one process at B = 32), with this rank's halves of the five
injected dropout masks, `random` seeded DIFFERENTLY on rank 1 --
  * the Gram matrix of the averaged task gradients, the PCGrad weights and the projection order == the reference's B = 32 step,
  * sampled post-step parameters / spectral-norm vectors == the reference's post-step state (same rule as
    tests/test_step_gpu.py::test_full_step_b32_vs_golden),
  * the 17 logged scalars, averaged over the two ranks, == the reference's (every logged term is a batch mean),
  * the generator's PSNR on all 32 patches after the step within 0.01 dB of the reference's,
  * both ranks end with bit-identical parameters.
The oracle module is used as the checker's fixture source only (seeded fill, synthetic batch), as in tests/test_step_gpu.py.
Exit code 0 = all of it held on both ranks.  This file is a script, not a pytest module: pytest's GPU process must not be
the one that starts GPU child processes."""
import json
import os
import random
import signal
import socket
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
TOL = 1e-3


def parent():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_RANK="0")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env0, RANK=str(r))) for r in range(2)]

    def stop(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        time.sleep(2.0)
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    rc, pending = 0, list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:
                    q.terminate()
        time.sleep(0.05)
    print(f"[dp2] ranks finished, exit code {rc}", flush=True)
    return rc


def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


def child(rank, world):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtdgan_oracle as orc
    import mtd_gan_amd  # noqa: F401
    from mtd_gan_amd import engine, parallel
    from mtd_gan_amd import metrics as M
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z2 = json.load(open(os.path.join(GOLD, "step_seeded.json")))
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z2["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z2["dfill"]).items()})
    m = MTD_GAN_Method()
    m.load_state_dict(full)
    m.to(dev).train()
    g = torch.Generator().manual_seed(z2["mask_seed"])
    masks = [(torch.rand(32, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    x, y = orc.synthetic_ldct(32, seed=z["data_seed"])
    per = 32 // world
    lo, hi = rank * per, (rank + 1) * per
    m.Discriminator._inject_masks = [k[lo:hi].clone() for k in masks]
    wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
    dp = parallel.DataParallelSync(dev)
    dp.broadcast_module(m)
    wm.method.dp = dp
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = FusedAdamW([dict(params=m.Discriminator.parameters(), lr=z["lr"], **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = FusedAdamW(m.Generator.parameters(), lr=z["lr"], **kw)
    random.seed(77 if rank == 0 else 1234567)           # rank 0 draws the reference's order; rank 1 would draw another
    captured = {}
    real_d_loss = m.d_loss

    def d_loss(a, b):
        out = real_d_loss(a, b)
        captured["tape"] = out[0]._mtd_tape
        return out
    m.d_loss = d_loss
    t0 = time.perf_counter()
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x[lo:hi], n_100=y[lo:hi])], oG, oD, dev, 0, 0, per, wm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tape = captured["tape"]
    fails = []

    def check(ok, what):
        if not ok:
            fails.append(what)
    # ---- projection order: rank 0's draw on every rank
    applied = tape.orders_dev.cpu().tolist()[:9]
    check(applied == [j for o in z["orders"] for j in o], f"applied order {applied} != reference {z['orders']}")
    if rank == 1:
        check(tape.orders != z["orders"], "rank 1 was meant to draw a different order of its own")
    # ---- Gram matrix / PCGrad weights of the AVERAGED task gradients against the reference's B = 32 step
    gram_ref = torch.tensor(z["gram"], dtype=torch.float64)
    scale = torch.sqrt(torch.outer(gram_ref.diag(), gram_ref.diag()))
    gerr = ((tape.gram.reshape(3, 3).cpu() - gram_ref).abs() / scale).max().item()
    check(gerr < TOL, f"gram {gerr}")
    werr = rel(tape.coeff[:3].cpu(), torch.tensor(z["pc_weights"]))
    check(werr < TOL, f"pcgrad weights {werr}")
    # ---- post-step state
    post = m.state_dict()
    bad = []
    for k, samples in z["post_samples"].items():
        t, t0_ = post[k].reshape(-1).cpu(), full[k].reshape(-1)
        for i, sv in enumerate(samples):
            idx = (i * 2654435761 + 12345) % t.numel()
            if k.endswith(("weight_u", "weight_v")):
                ok = abs(t[idx].item() - sv) <= TOL * max(abs(sv), 1e-3)
            else:
                ok = abs((t[idx].item() - t0_[idx].item()) - (sv - t0_[idx].item())) <= 0.05 * z["lr"] + 1e-9
            if not ok:
                bad.append((k, i))
    check(len(bad) <= 0.01 * 4 * len(z["post_samples"]), f"{len(bad)} post-step samples off: {bad[:6]}")
    check(torch.equal(post["Discriminator.c_fc.weight_orig"].cpu(), full["Discriminator.c_fc.weight_orig"]), "c_fc moved")
    # ---- logged scalars: mean over the two shards == the reference's batch means
    keys = list(z["stats"].keys())
    sv = torch.tensor([stats[k] for k in keys], dtype=torch.float64)
    dist.all_reduce(sv)
    sv /= world
    serr = 0.0
    for k, v in zip(keys, sv.tolist()):
        want = z["stats"][k]
        serr = max(serr, abs(v - want) / (abs(want) + 2e-3))
        check(abs(v - want) <= TOL * abs(want) + 2e-6, f"stat {k}: {v} vs {want}")
    # ---- generator after the step, on the whole batch
    with torch.no_grad():
        pred = m.Generator(x.to(dev))
    psnr = float(M.compute_PSNR(x.to(dev), y.to(dev), pred.clip(0, 1))[2])
    check(abs(psnr - z["post_metrics"]["psnr"]) <= 0.01, f"psnr {psnr} vs {z['post_metrics']['psnr']}")
    # ---- both ranks hold the same bits
    h = torch.zeros(2, dtype=torch.float64)
    for k, v in post.items():
        if v.is_floating_point():
            vv = v.double().reshape(-1)
            h[0] += vv.sum().cpu()
            h[1] += (vv * vv).sum().cpu()
    both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, h)
    check(all(torch.equal(both[0], b) for b in both), f"ranks diverged: {[b.tolist() for b in both]}")
    # ---- round 4: the RECORDED launch list under data parallelism.  Five more iterations per rank on fresh models, once with
    # every iteration eager and once as engine.train_MTD_GAN_Ours runs them by default (two eager, one recorded, two replayed:
    # the collectives, the broadcast of rank 0's projection order and the stream hand-offs are part of the list): the two
    # must agree BIT FOR BIT on every rank (dropout on, drawn from torch's generator), and the ranks with each other.
    from mtd_gan_amd import train_step as TS
    finals = {}
    for mode in ("eager", "list"):
        mm = MTD_GAN_Method()
        mm.load_state_dict(full)
        mm.to(dev).train()
        wm2 = WeightMethods("pcgrad", n_tasks=3, device=dev)
        dp2 = parallel.DataParallelSync(dev)
        dp2.broadcast_module(mm)
        wm2.method.dp = dp2
        oD2 = FusedAdamW(mm.Discriminator.parameters(), lr=z["lr"], **kw)
        oG2 = FusedAdamW(mm.Generator.parameters(), lr=z["lr"], **kw)
        random.seed(5 + 100 * rank)
        torch.manual_seed(900 + rank)
        TS.LIST_MODE = mode == "list"
        loader = [dict(n_20=x[lo:hi].roll(i, 0), n_100=y[lo:hi].roll(i, 0)) for i in range(5)]
        st5 = engine.train_MTD_GAN_Ours(mm, loader, oG2, oD2, dev, 0, 0, per, wm2)
        torch.cuda.synchronize()
        if mode == "list":
            check(isinstance(getattr(mm, "_mtd_recorded", None), TS.RecordedTrainStep) and mm._mtd_recorded.iterations == 3,      # (the recorded iteration + two replays)
                  f"list mode did not replay: {getattr(mm, '_mtd_recorded', None)}")
        finals[mode] = ({k: v.clone() for k, v in mm.state_dict().items()}, st5)
        del mm, oD2, oG2
    TS.LIST_MODE = True
    check(finals["eager"][1] == finals["list"][1], f"5-iteration stats differ: {finals['eager'][1]} vs {finals['list'][1]}")
    ndiff = sum(0 if torch.equal(finals["eager"][0][k], finals["list"][0][k]) else 1 for k in finals["eager"][0])
    check(ndiff == 0, f"{ndiff} tensors differ between eager and replayed iterations")
    h5 = torch.zeros(2, dtype=torch.float64)
    for k, v in finals["list"][0].items():
        if v.is_floating_point():
            vv = v.double().reshape(-1)
            h5[0] += vv.sum().cpu()
            h5[1] += (vv * vv).sum().cpu()
    both5 = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both5, h5)
    check(all(torch.equal(both5[0], b) for b in both5), f"ranks diverged after replays: {[b.tolist() for b in both5]}")
    print(f"[dp2 rank {rank}] recorded-list rehearsal: 5 iterations (2 eager + 1 recorded + 2 replayed) == 5 eager, {ndiff} tensors differ; "
          f"ranks identical: {all(torch.equal(both5[0], b) for b in both5)}", flush=True)
    print(f"[dp2 rank {rank}] step {dt * 1e3:.0f} ms (first step, gloo through the host); gram err {gerr:.2e}, pcgrad-weight err {werr:.2e}, "
          f"worst stat err {serr:.2e}, post-step samples off {len(bad)}/{4 * len(z['post_samples'])}, PSNR {psnr:.4f} dB "
          f"(reference {z['post_metrics']['psnr']:.4f}); {'OK' if not fails else 'FAILED: ' + '; '.join(fails)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 1 if fails else 0


if __name__ == "__main__":
    if "RANK" in os.environ:
        sys.exit(child(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])))
    sys.exit(parent())
