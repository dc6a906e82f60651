import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mtd_gan_amd import engine
from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
from mtd_gan_amd.module.weight_methods import WeightMethods
from mtd_gan_amd.optimizers import FusedAdamW
from mtd_gan_amd.train_step import GraphedTrainStep
from mtd_gan_amd.data import synthetic_ldct
x, y = synthetic_ldct(4, seed=77)
x, y = x.cuda(), y.cuda()

def make():
    torch.manual_seed(5)
    m = MTD_GAN_Method().cuda().train()
    m.Discriminator.c_drop.p = 0.0
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
    oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
    random.seed(123)
    return m, wm, oD, oG

for tag in (os.environ.get("EAGER_RUNS", "eager A,eager B").split(",") if os.environ.get("EAGER_RUNS", "x") else []):
    m, wm, oD, oG = make()
    out = []
    for _ in range(5):
        names, vals = engine.train_iteration(m, x, y, oG, oD, wm, None)
        out.append(vals[:2].tolist() + [vals[11].item()])
    print(tag, out, flush=True)
m, wm, oD, oG = make()
out = []
class Rec:
    pass
import mtd_gan_amd.train_step as TS
orig = engine.train_iteration
def rec_iter(*a, **k):
    n, v = orig(*a, **k)
    return n, v
gs = GraphedTrainStep(m, oG, oD, wm, x, y, warmup=3)
for _ in range(2):
    names, vals = gs.step()
    torch.cuda.synchronize()
    out.append(vals[:2].tolist() + [vals[11].item()])
print("graph (steps 4,5)", out, flush=True)
print("adam steps", oD.state[m.Discriminator.enc_out.weight]["step"], oG.state[m.Generator.encoder[0].weight]["step"])
