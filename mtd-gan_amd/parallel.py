"""Batch-sharded data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over
xGMI.  Replaces the reference's single-process nn.DataParallel (train.py:93-98), which cannot run the
PCGrad path at all.  To reproduce the large-batch step exactly, PCGrad is nonlinear in
the gradients, so each per-task shared-gradient vector is averaged across ranks BEFORE the projection:
three 114 MB all-reduces per iteration, each enqueued on a side stream as soon as that task's backward
has been issued so that it overlaps the next task's backward kernels (the last task ships its finished
slices while it still runs: train_step.DStepTape.run_pcgrad); rank 0's shuffle order is broadcast.  On CPU the same code runs on the gloo backend (tests)."""
from . import _options
import torch
import torch.distributed as dist


class DataParallelSync:
    def __init__(self, device=None):
        if not dist.is_initialized():
            raise RuntimeError("DataParallelSync needs an initialised torch.distributed process group")
        self.world = dist.get_world_size()
        # MTD_FORCE_DP=1: run every collective even in a one-rank group (single-GPU smoke test of the N > 1 code path)
        import os
        self.force = _options.product("MTD_FORCE_DP", "0") == "1"
        self.device = device
        self.cuda = device is not None and torch.device(device).type == "cuda"
        # RCCL averages inside the collective (ReduceOp.AVG); gloo -- CPU tests, and the two-ranks-on-one-GPU rehearsal of
        # tests/dp_two_ranks_one_gpu.py, where RCCL cannot put two ranks on one device -- only sums: scale afterwards
        self.has_avg = dist.get_backend() == "nccl"
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self._pending = False
        # stub = True: every data-path collective (all-reduce of gradients, broadcast of the projection order) becomes a no-op AT CALL
        # TIME -- the stream hand-offs around it stay, and so does a recorded launch list (its entries are these closures).  A
        # measurement switch: bench.py's `comm_exposed_ms` is (step) - (step with stub).  The replicas drift apart while it is on.
        self.stub = False

    def _coll(self, fn):
        """A data-path collective as a closure that honours `stub` when it RUNS (also when a launch list replays it)."""
        def run():
            if not self.stub:
                fn()
        run.mtd_collective = True      # (tests look for these entries in a recorded launch list)
        return run

    def broadcast_module(self, module):
        """Rank 0's parameters and buffers to every rank.  The weights change under every cache of derived quantities
        (packed conv weights, transposed mix weights, the generator tape), so those are dropped and the parameters'
        `_mtd_epoch` stamps bumped: a forward that ran before the broadcast must not leave stale views behind."""
        from . import kernels as K
        params = list(module.parameters())
        with torch.no_grad():
            for t in params + list(module.buffers()):
                dist.broadcast(t.detach(), src=0)          # shares the version counter with t (unlike t.data)
        K.weights_changed(params)
        K.weights_changed(None)

    def broadcast_optimizer_state(self, optimizers):
        """Rank 0's AdamW moment tensors (exp_avg, exp_avg_sq) and step counters to every rank, in place: the companion of
        broadcast_module when replicas have drifted (train_step.recorded_iteration) -- moments that were updated from diverged
        gradients would pull the re-broadcast weights apart again on the next step."""
        with torch.no_grad():
            for opt in optimizers:
                for group in opt.param_groups:
                    for p in group["params"]:
                        st = opt.state.get(p)
                        if not st:
                            continue
                        for key in ("exp_avg", "exp_avg_sq", "step"):
                            t = st.get(key)
                            if torch.is_tensor(t):
                                dist.broadcast(t, src=0)

    def broadcast_orders(self, orders, device=None, upload=None):
        """The PCGrad projection order of this step as ONE collective decision: rank 0's draw wins.  Every rank still
        consumes its own `random` stream exactly like the reference (module/weight_methods.py:452), so a single process
        behaves as before, but ranks whose `random` states differ (seed+rank conventions, data loaders consuming
        `random`) can no longer apply different projections to the same averaged gradients.
        orders: list of T index lists.  Returns an int32 tensor (T*T, padded to 16) on `device` holding rank 0's order;
        on GPUs the broadcast runs on the collective side stream (no host synchronisation) and wait() joins it.
        upload: optional callable(list_of_ints) -> int32 device tensor (train_step supplies a pinned-slot upload so that
        the step has no hipMemcpy); the default builds the tensor with torch."""
        flat = [j for o in orders for j in o]
        flat = flat + [0] * (16 - len(flat))
        if self.cuda:
            t = upload(flat) if upload is not None else torch.tensor(flat, dtype=torch.int32).to(device or self.device)
            if self.world > 1 or self.force:
                self._on_side(self._coll(lambda: dist.broadcast(t, src=0)), t)
            return t
        t = torch.tensor(flat, dtype=torch.int32)
        if self.world > 1 and not self.stub:
            dist.broadcast(t, src=0)
        return t

    def _on_side(self, collective, tensor, after=()):
        """Enqueue `collective` on the RCCL side stream, ordered after everything enqueued so far on the current stream and on
        the streams in `after`.  Stream order and the call itself go through kernels.order_streams / kernels.rec, so a
        kernels.LaunchList that is recording the step records them too (a replay re-issues the collective on the same tensor)."""
        from . import kernels as K
        K.order_streams(self.side, torch.cuda.current_stream())
        for st in after:
            K.order_streams(self.side, st)
        tensor.record_stream(self.side)
        with torch.cuda.stream(self.side):
            K.rec(collective)
        self._pending = True

    def all_reduce_avg_inline(self, flat):
        """Average a SMALL tensor across ranks on the CURRENT stream (the generator's 1.9 MB of gradients, needed by the very
        next launch): no side stream, no event hand-offs -- at this size the two cross-stream hops cost more than the collective."""
        if self.world == 1 and not self.force:
            return
        if not self.cuda:
            return self.all_reduce_avg(flat)
        from . import kernels as K
        self.wait()                                   # (collectives on the side stream that may touch the same communicator: keep issue order)
        if self.has_avg:
            op = dist.ReduceOp.AVG if self.world > 1 else dist.ReduceOp.SUM
            K.rec(self._coll(lambda: dist.all_reduce(flat, op=op)))
        else:
            def sum_and_scale():
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.mul_(1.0 / self.world)
            K.rec(self._coll(sum_and_scale))

    def all_reduce_avg(self, flat, after=()):
        """Average `flat` (a contiguous tensor) across ranks.  On GPUs the collective runs on a side stream
        ordered after the kernels already enqueued on the current stream and on the streams in `after` (the weight-gradient
        side stream that fills a task vector: the collective waits for it, the main stream does not have to); wait() joins
        the collectives back."""
        if self.world == 1 and not self.force:
            return
        if self.cuda:
            if self.has_avg:
                # RCCL scales inside the collective: no extra pass over 114 MB.  (A ONE-rank group -- MTD_FORCE_DP=1, the
                # plumbing rehearsal -- asks for SUM, which equals AVG there: RCCL runs a one-rank AVG as a pre-multiply kernel
                # that rewrites the whole buffer in place, 112 us per 114 MB, which no multi-rank run executes.)
                op = dist.ReduceOp.AVG if self.world > 1 else dist.ReduceOp.SUM
                self._on_side(self._coll(lambda: dist.all_reduce(flat, op=op)), flat, after)
            else:
                def sum_and_scale():
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                    flat.mul_(1.0 / self.world)
                self._on_side(self._coll(sum_and_scale), flat, after)
            # consumers of `flat` are enqueued later on the current stream, after wait(): the overlap window is the backward
            # kernels issued before the next consumer (PCGrad's Gram kernel is the first, after all three tasks)
        elif not self.stub:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.world)

    def wait(self):
        """The current stream waits for every collective enqueued so far."""
        if self.cuda and self._pending:
            from . import kernels as K
            K.order_streams(torch.cuda.current_stream(), self.side)
        self._pending = False

    def all_agree(self, ok):
        """One MIN all-reduce of a flag: True only if EVERY rank passed True.  Ranks decide together whether the recorded launch
        list of an iteration is used (train_step.recorded_iteration): a rank whose recording failed must not leave the others
        replaying while it runs eagerly with a different schedule of stream hand-offs.  Synchronises the host once."""
        if self.world == 1 and not self.force:
            return bool(ok)
        self.wait()
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if self.cuda else None)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def replicas_agree(self, tensors):
        """Do all ranks hold bit-identical copies of `tensors` (contiguous fp32)?  The invariant of synchronous data
        parallelism -- the reference gets it from nn.DataParallel's single set of weights (train.py:93-98) -- and what any
        missed ordering edge around a collective would break first: a rank that reads a gradient buffer before its all-reduce
        has landed updates with its own shard's gradient.  Integer checksums (mtd_checksum_multi) compared by a MIN and a MAX
        all-reduce; synchronises the host once."""
        if self.world == 1 and not self.force:
            return True
        tensors = [t.detach() for t in tensors]
        if self.cuda:
            from . import kernels as K
            sums = K.checksum_multi(tensors)
        else:
            sums = torch.stack([t.reshape(-1).view(torch.int32).to(torch.int64).sum() for t in tensors])
        self.wait()
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    def all_reduce_avg_list(self, tensors):
        """Bucket a list of tensors into one flat buffer, average, scatter back."""
        if (self.world == 1 and not self.force) or self.stub:
            return
        self.wait()
        flat = torch.cat([t.reshape(-1) for t in tensors])
        if self.cuda and self.has_avg:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)          # gloo has no AVG
            flat.mul_(1.0 / self.world)
        ofs = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[ofs:ofs + n].view_as(t))
            ofs += n
