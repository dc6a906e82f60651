import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import mtd_gan_amd  # noqa
from mtd_gan_amd.train_step import FullStepWorkload
wl = FullStepWorkload(torch.device("cuda", 0), 0, 1, 32)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
def loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("bench loop 20:", loop(20), " 60:", loop(60))
print("api 20:", wl.engine_api_ms(20), " 60:", wl.engine_api_ms(60), " 20 again:", wl.engine_api_ms(20))
print("bench loop 20:", loop(20))
