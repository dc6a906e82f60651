"""Workloads timed by bench.py.  Each exposes step() (one pass of the hot path over one resident batch),
config(), extra() and cpu_baseline_spec()."""
from . import _options
import torch
import torch.distributed as dist

from .data import synthetic_ldct


class GeneratorWorkload:
    """BASELINE config 2: generator-only Res-FFT-Conv forward + backward, 32 x 1 x 64 x 64 fp32 per GPU.
    Algorithmic work 10.72 GFLOP/patch."""

    name = "generator_fwd_bwd"
    gflop_per_patch = 10.72

    def __init__(self, dev, rank, world, batch):
        from .arch.Ours.networks import ResFFT_Generator
        torch.manual_seed(2024)
        self.G = ResFFT_Generator(1, 32, 10, 3, 1).to(dev)
        x, y = synthetic_ldct(batch, seed=1234 + rank)
        self.x, self.y = x.to(dev), y.to(dev)
        self.world, self.batch = world, batch
        self.params = [p for p in self.G.parameters()]
        self.flat = None
        self.graph = None
        self.graph_error = None
        self.launch_list = None
        import os
        mode = _options.product("MTD_GRAPH", "1")
        if world == 1 and mode == "list":
            # recorded launch list (kernels.LaunchList): the step's C-ABI calls and stream-order operations re-issued without
            # the Python around them, side streams kept.  For THIS workload the captured single-stream graph is the faster
            # replay (round 2, docs/LAB_NOTES_r1_r2.md: the kernels of a Res-FFT block cannot share a CU -- the halo-tile conv
            # holds 136 KB of LDS, one workgroup per CU -- so a second stream only adds event waits between them); the full
            # training iteration, whose weight gradients do overlap its data-gradient chain, replays as a list by default
            # (train_step.RecordedTrainStep).
            from . import kernels as K
            try:
                for _ in range(2):
                    self.step_eager()
                ll = K.LaunchList()
                ll.record(self.step_eager, dev)
                self.launch_list = ll
                self.list_grads = [p.grad for p in self.params]
            except Exception as e:
                self.graph_error = repr(e)
                self.launch_list = None
                torch.cuda.synchronize()
        if world == 1 and mode == "1":
            # hipGraph replay of the forward + backward (static shapes, no host-side state): eager launches of this workload
            # are host-bound (the Python enqueue of its ~430 launches takes longer than the 5.4 ms of GPU work).  Captured
            # with every kernel in ONE stream: a single-stream graph replays at the kernels' own pace, a captured multi-stream
            # section does not (ROCm 7.2 serialises it with extra waits).  MTD_GRAPH=0 keeps the eager launches.
            from . import kernels as K
            try:
                K.set_concurrency(False)
                K.prepare_capture(dev)
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        self.step_eager()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                K.CAPTURE_TAG += 1
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.step_eager()
                self.graph = g
            except Exception as e:
                self.graph_error = repr(e)
                self.graph = None
                torch.cuda.synchronize()
            finally:
                K.set_concurrency(True)

    def step(self):
        if self.launch_list is not None:
            self.launch_list.replay()
            if self.params[0].grad is not self.list_grads[0]:
                for p, g in zip(self.params, self.list_grads):
                    p.grad = g
        elif self.graph is not None:
            self.graph.replay()
        else:
            self.step_eager()

    def step_eager(self):
        for p in self.params:
            p.grad = None
        out = self.G(self.x)
        out.backward(self.y)          # cotangent = fixed tensor (out.mean().backward() semantics up to scale)
        if self.world > 1:
            flat = torch.cat([p.grad.reshape(-1) for p in self.params])
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            self.flat = flat

    def config(self, world):
        return {"workload": "Generator-only Res-FFT-Conv fwd+bwd (BASELINE configs[1])", "per_gpu_batch": self.batch,
                "global_batch": self.batch * world, "patch": "1x64x64", "parallelism": f"dp{world}"}

    def extra(self):
        return {"algorithmic_gflop_per_patch": self.gflop_per_patch,
                "launch_mode": ("launch-list replay, side streams" if self.launch_list is not None else
                                "hipGraph replay" if self.graph is not None else "eager launches"), "graph_error": self.graph_error}


class InferenceWorkload:
    """BASELINE configs[4]: whole-slice generator inference on (8,1,512,512) per GPU, torch.no_grad (what the reference's
    valid_/test_MTD_GAN_Ours do, engine.py:89,129), followed by the pixel metrics of the test loop.  Work: 64 x 3.63 =
    232 GFLOP per slice.  A "step" is one batch of 8 slices."""

    name = "inference512"
    gflop_per_patch = 232.0
    slices = 8

    def __init__(self, dev, rank, world, batch):
        from .arch.Ours.networks import ResFFT_Generator
        torch.manual_seed(2024)
        self.G = ResFFT_Generator(1, 32, 10, 3, 1).to(dev).eval()
        x, y = synthetic_ldct(self.slices, seed=1234 + rank, size=512)
        self.x, self.y = x.to(dev), y.to(dev)
        self.world = world
        self.last = None

    def step(self):
        from . import kernels as K
        from . import metrics as M
        with torch.no_grad():
            pred = self.G(self.x)
            self.last = M.pixel_metrics(self.x, self.y, K.clip01(pred))

    def config(self, world):
        return {"workload": "Whole-slice generator inference + pixel metrics (BASELINE configs[4])", "per_gpu_batch": self.slices,
                "global_batch": self.slices * world, "patch": "1x512x512", "parallelism": f"dp{world}"}

    def extra(self):
        return {"algorithmic_gflop_per_patch": self.gflop_per_patch, "unit_note": "img = one 512x512 slice",
                "pixel_metrics_last_batch": self.last}


class PatchSamplerWorkload:
    """the training-patch front end (create_datasets/Mayo.py:117-136) on the device.  A "step" produces one
    training batch: 32 patch pairs (8 samples from each of 4 resident 512 x 512 HU slice pairs), fresh random choices every
    step (drawn on the host, uploaded as a descriptor table)."""

    name = "patch_sampler"
    n_slices = 4
    dtype = "i16 -> f32"

    def __init__(self, dev, rank, world, batch):
        import numpy as np
        from .create_datasets import Mayo
        self.Mayo = Mayo
        lo, hi = Mayo.synthetic_hu_slices(self.n_slices, size=512, seed=77 + rank)
        self.lo, self.hi = lo.to(dev), hi.to(dev)
        self.rng = np.random.RandomState(5 + rank)
        self.world, self.batch = world, batch
        self.last = None
        # bytes a step has to move: the box pass reads the full-dose slices, the gather reads 2 bytes and writes 4 per pixel and dose
        self.roofline_bytes_per_step = self.n_slices * 512 * 512 * 2 + 2 * 32 * 64 * 64 * (2 + 4)

    def step(self):
        d = self.Mayo.draw_descriptors(self.rng, self.n_slices)
        self.last = self.Mayo.window_patches(self.lo, self.hi, d)

    def config(self, world):
        return {"workload": "Training-patch front end: HU window + foreground crop + 8 x 64x64 samples per slice + rot90 / flip / "
                            "rotate (Mayo.py window_patch), 4 slices per step", "per_gpu_batch": 32, "global_batch": 32 * world,
                "patch": "1x64x64 from 512x512 int16 HU", "parallelism": f"dp{world}"}

    def extra(self):
        return {"unit_note": "img = one (quarter dose, full dose) patch pair"}


def make(name, dev, rank, world, batch):
    if name in ("patch_sampler", "sampler"):
        return PatchSamplerWorkload(dev, rank, world, batch)
    if name in ("inference", "inference512"):
        return InferenceWorkload(dev, rank, world, batch)
    if name in ("auto", "generator", "generator_fwd_bwd"):
        try:
            if name == "auto":
                from .train_step import FullStepWorkload   # becomes the default once the full step exists
                return FullStepWorkload(dev, rank, world, batch)
        except ImportError:
            pass
        return GeneratorWorkload(dev, rank, world, batch)
    if name == "full_step":
        from .train_step import FullStepWorkload
        return FullStepWorkload(dev, rank, world, batch)
    raise ValueError(f"unknown workload {name}")
