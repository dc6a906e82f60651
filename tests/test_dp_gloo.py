"""CPU test of the N > 1 path: world_size-2 gloo run of parallel.DataParallelSync with the data-parallel
PCGrad semantics (average each per-task shared-gradient vector across ranks BEFORE the projection; same
shuffle order on every rank).  "2 shards + all-reduce" must equal the single-process large-batch step
(SURVEY 8e).  The arithmetic on each rank is the CPU oracle (this is a test); the collective logic is the
product code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mtdgan_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _task_grads(state, x, y, masks):
    shared = ["Discriminator." + n for n in orc.d_shared_names()]
    tspec = ["Discriminator." + n for n in orc.d_task_specific_names()]
    st = {k: v.clone() for k, v in state.items()}
    for n in shared + tspec:
        st[n] = st[n].requires_grad_(True)
    lo, _, _ = orc.d_loss(st, x, y, masks)
    sp = [st[n] for n in shared]
    flat = [torch.cat([g.reshape(-1) for g in torch.autograd.grad(lo[i], sp, retain_graph=True)]) for i in range(3)]
    ts = torch.autograd.grad(lo.sum(), [st[n] for n in tspec])
    return lo.detach(), flat, torch.cat([g.reshape(-1) for g in ts])


def _merge(flat, orders):
    gram = [[float(torch.dot(a.double(), b.double())) for b in flat] for a in flat]
    w = orc.pcgrad_coefficients(gram, orders)
    return sum(wk * f for wk, f in zip(w, flat))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mtd_gan_amd.parallel import DataParallelSync
    dp = DataParallelSync(device=None)
    state = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items()}
    state.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=9).items()})
    x, y = orc.synthetic_ldct(world, seed=1234)
    g = torch.Generator().manual_seed(33)
    masks = [(torch.rand(world, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(4)]
    orders = [[2, 0, 1], [1, 2, 0], [0, 2, 1]]
    lo, flat, ts = _task_grads(state, x[rank:rank + 1], y[rank:rank + 1], [m[rank:rank + 1] for m in masks])
    for f in flat:
        dp.all_reduce_avg(f)
    dp.wait()
    tsl = [ts]
    dp.all_reduce_avg_list(tsl)
    merged = _merge(flat, orders)
    lo_avg = lo.clone()
    dp.all_reduce_avg(lo_avg)
    if rank == 0:
        big_lo, big_flat, big_ts = _task_grads(state, x, y, masks)
        big_merged = _merge(big_flat, orders)
        rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
        out.put(dict(loss=rel(lo_avg, big_lo), merged=rel(merged, big_merged), ts=rel(tsl[0], big_ts),
                     tasks=[rel(a, b) for a, b in zip(flat, big_flat)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_shards_plus_allreduce_equal_large_batch():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(540)
        assert p.exitcode == 0
    res = q.get()
    assert res["loss"] < 1e-5, res
    assert max(res["tasks"]) < 5e-3, res          # fp32 sums in a different order; consist gradients are tiny
    assert res["merged"] < 5e-3 and res["ts"] < 5e-3, res


def _orders_worker(rank, world, port, out):
    """Ranks whose global `random` states differ (the seed+rank convention) must still apply ONE projection order:
    rank 0's draw, taken from its own `random` stream exactly as the reference consumes it."""
    import random
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mtd_gan_amd import train_step as TS
    from mtd_gan_amd.parallel import DataParallelSync
    dp = DataParallelSync(device=None)
    random.seed(100 + rank)
    got = []
    for _ in range(3):
        mine = TS.next_orders(3)
        got.append((mine, dp.broadcast_orders(mine).tolist()))
    out.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_projection_order_is_a_collective_decision():
    import random
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_orders_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get() for _ in procs)
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    random.seed(100)
    expect = [[j for o in orc.shuffle_orders(3) for j in o] for _ in range(3)]
    for step in range(3):
        for rank in range(2):
            assert res[rank][step][1][:9] == expect[step], (rank, step, res[rank][step])
    assert any(res[1][s][0] != res[0][s][0] for s in range(3))        # the ranks' own draws did differ


def _ablation_worker(rank, world, port, out):
    """engine.train_MTD_GAN_Ours on the ablation wrappers' path (method_D=None: no weight method to carry the data-parallel
    hook) in a two-rank group: without a hook it must refuse to train unsynchronised replicas; with dp= it averages the
    discriminator and generator gradients across the ranks (a tiny CPU stand-in for the wrapper: the collective logic and
    the engine's control flow are the product's, no kernel runs)."""
    import torch.nn as nn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mtd_gan_amd import engine
    from mtd_gan_amd.parallel import DataParallelSync

    class Wrapper(nn.Module):                       # an Ablation_* look-alike: d_loss / g_loss return one scalar + a dict
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.Generator, self.Discriminator = nn.Linear(4, 4), nn.Linear(4, 1)

        def d_loss(self, x, y):
            lo = (self.Discriminator(y) - 1).pow(2).mean() + self.Discriminator(self.Generator(x).detach()).pow(2).mean()
            return lo, {"D/a": lo.detach()}

        def g_loss(self, x, y):
            lo = (self.Generator(x) - y).abs().mean()
            return lo, {"G/a": lo.detach()}
    m = Wrapper()
    oG, oD = torch.optim.SGD(m.Generator.parameters(), lr=0.1), torch.optim.SGD(m.Discriminator.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(10 + rank)       # every rank its own shard
    loader = [dict(n_20=torch.randn(8, 4, generator=g), n_100=torch.randn(8, 4, generator=g))]
    refused = False
    try:
        engine.train_MTD_GAN_Ours(m, loader, oG, oD, torch.device("cpu"), 0, 0, 8, None)
    except RuntimeError as e:
        refused = "data-parallel hook" in str(e)
    engine.train_MTD_GAN_Ours(m, loader, oG, oD, torch.device("cpu"), 0, 0, 8, None, dp=DataParallelSync(device=None))
    w = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    both = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    out.put((rank, refused, bool(torch.equal(both[0], both[1]))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ablation_path_needs_and_uses_the_dp_hook():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_ablation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for rank, refused, same in res:
        assert refused, rank          # no hook in a two-rank group: refused
        assert same, rank             # with the hook: the replicas stay identical (gradients averaged before each update)


def _agree_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mtd_gan_amd.parallel import DataParallelSync
    dp = DataParallelSync(device=None)
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(1000, generator=g), torch.randn(33, generator=g)
    res = dict(same=dp.replicas_agree([a, b]))
    b2 = b.clone()
    if rank == 1:
        b2.view(torch.int32)[7] ^= 1                 # one bit on one rank
    res["one_bit"] = dp.replicas_agree([a, b2])
    res["all_ok"] = dp.all_agree(True)
    res["one_failed"] = dp.all_agree(rank != 1)      # rank 1's recording "failed": every rank must hear of it
    # the repair of drifted replicas carries rank 0's AdamW moments with its weights (advisor, round 5): broadcast_optimizer_state
    p = torch.nn.Parameter(a.clone())
    p.grad = torch.full_like(p, 1.0 + rank)          # diverged gradients -> diverged moments
    opt = torch.optim.AdamW([p], lr=1e-3)
    opt.step()
    m = opt.state[p]["exp_avg"]
    res["moments_differ"] = not dp.replicas_agree([m])
    dp.broadcast_optimizer_state([opt])
    res["moments_repaired"] = dp.replicas_agree([m, opt.state[p]["exp_avg_sq"]])
    # stub: a measurement switch (bench.py comm_exposed_ms) -- the data-path collectives become no-ops when they run
    v = torch.full((4,), float(rank))
    dp.stub = True
    dp.all_reduce_avg(v)
    res["stub_is_a_no_op"] = bool((v == float(rank)).all())
    dp.stub = False
    dp.all_reduce_avg(v)
    res["live_again"] = bool((v == 0.5).all())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_decide_together_and_replica_divergence_is_seen():
    """Round 5 (the first N > 1 run must be boring): the flag all-reduce by which ranks agree to use / drop the recorded launch
    list, and the comparison of replicas that guards the first data-parallel replays, at world size 2 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get() for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in (0, 1):
        assert got[r] == dict(same=True, one_bit=False, all_ok=True, one_failed=False, moments_differ=True, moments_repaired=True,
                              stub_is_a_no_op=True, live_again=True), (r, got[r])
