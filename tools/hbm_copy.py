"""Stream-copy microbenchmark: achieved HBM bandwidth of a plain device-to-device copy and of a read-only reduction,
next to the 8 TB/s spec figure the HBM-bound kernels are priced against.  python tools/hbm_copy.py"""
import torch

dev = torch.device("cuda", 0)
for mib in (256, 1024, 4096):
    n = mib * 2**20 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * n * 4), ("sum  (read only)", lambda: a.sum(), n * 4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mib:5d} MiB  {name:20s} {ms * 1e3:8.1f} us  {bytes_moved / ms / 1e9 * 1e3 / 1e3:7.2f} TB/s")
