"""Batch-sharded data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over
xGMI.  Replaces the reference's single-process nn.DataParallel (train.py:93-98), which cannot run the
PCGrad path at all.  To reproduce the large-batch step exactly, PCGrad is nonlinear in
the gradients, so each per-task shared-gradient vector is averaged across ranks BEFORE the projection:
three 114 MB all-reduces per iteration, each enqueued on a side stream as soon as that task's backward
has been issued so that it overlaps the next task's backward kernels; all ranks draw the same shuffle
order (same `random` seed).  On CPU the same code runs on the gloo backend (tests)."""
import torch
import torch.distributed as dist


class DataParallelSync:
    def __init__(self, device=None):
        if not dist.is_initialized():
            raise RuntimeError("DataParallelSync needs an initialised torch.distributed process group")
        self.world = dist.get_world_size()
        # MTD_FORCE_DP=1: run every collective even in a one-rank group (single-GPU smoke test of the N > 1 code path)
        import os
        self.force = os.environ.get("MTD_FORCE_DP", "0") == "1"
        self.device = device
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self._pending = []

    def broadcast_module(self, module):
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0)

    def all_reduce_avg(self, flat):
        """Average `flat` (a contiguous tensor) across ranks.  On GPUs the collective runs on a side stream
        ordered after the kernels already enqueued on the current stream; wait() joins it back."""
        if self.world == 1 and not self.force:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            flat.record_stream(self.side)
            with torch.cuda.stream(self.side):
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)      # RCCL scales inside the collective: no extra pass over 114 MB
            done = torch.cuda.Event()
            done.record(self.side)
            self._pending.append(done)
            # consumers of `flat` are enqueued later on the current stream: make it wait right away for
            # correctness; the overlap window is the backward kernels issued before the next consumer.
            # (PCGrad's Gram kernel is the first consumer, after all three tasks.)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.world)

    def wait(self):
        if self.cuda:
            cur = torch.cuda.current_stream()
            for ev in self._pending:
                cur.wait_event(ev)
        self._pending = []

    def all_reduce_avg_list(self, tensors):
        """Bucket a list of tensors into one flat buffer, average, scatter back."""
        if self.world == 1 and not self.force:
            return
        self.wait()
        flat = torch.cat([t.reshape(-1) for t in tensors])
        if self.cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)          # gloo has no AVG
            flat.mul_(1.0 / self.world)
        ofs = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[ofs:ofs + n].view_as(t))
            ofs += n
