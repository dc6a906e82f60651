"""Every run-time switch of the package, in one place (round 5: "freeze the switchboard").

PRODUCT switches -- read from the environment by a shipped installation, each exercised by a test:

  MTD_LIST=0          engine.train_MTD_GAN_Ours / bench.py keep every iteration eager (default 1: two eager iterations, the
                      third recorded as a launch list, later ones replayed)
  MTD_LIST_DP=0       ... only under data parallelism (default 1)
  MTD_FORCE_DP=1      run every collective and stream hand-off of the N > 1 path in a one-rank group (rehearsal;
                      tests/test_step_gpu.py::test_data_parallel_path_on_one_rank_equals_plain_step)
  MTD_GRAPH=1         bench.py: replay the step as a captured hipGraph instead (slower on ROCm 7.2; tests/test_generator_gpu.py)
  MTD_GC_FREEZE=1     engine.train_MTD_GAN_Ours calls gc.freeze() once (process-wide)
  MTD_DP_SHARE_GPU=1  bench.py --gpus N with all ranks on device 0 over gloo (plumbing rehearsal)
  MTD_BENCH_WORKLOAD  bench.py's default --workload
  MTD_LAB=1           master switch of everything below
(the reading of each: tests/test_host_cpu.py::test_product_switches_are_read_from_the_environment)

LAB switches -- the kernel-selection / ablation variables of rounds 1-4 (MTD_NO_*, MTD_WINOGRAD*, MTD_FIRST_WRITE, ...) are read
ONLY when MTD_LAB=1 is set; without it a stray variable in a user's shell changes nothing.  The native library's own
switches need a lab build on top (`MTD_LAB_BUILD=1 python mtd-gan_amd/_build.py`, which writes libmtdgan_hip_lab.so; it is
loaded instead of the shipped library when MTD_LAB=1, it exists, it is not older than the kernel sources, and MTD_LAB_LIB=0 does
not ask for the shipped library with the Python-level lab switches only -- bench.py's PMC child processes do).  Tests flip module
attributes, not the environment."""
import os

LAB = os.environ.get("MTD_LAB", "0") == "1"
PRODUCT = ("MTD_LIST", "MTD_LIST_DP", "MTD_FORCE_DP", "MTD_GRAPH", "MTD_GC_FREEZE", "MTD_DP_SHARE_GPU", "MTD_BENCH_WORKLOAD", "MTD_LAB")


def product(name, default):
    assert name in PRODUCT, name
    return os.environ.get(name, default)


def lab(name, default):
    """A lab switch: the environment's value under MTD_LAB=1, the default otherwise."""
    return os.environ.get(name, default) if LAB else default
