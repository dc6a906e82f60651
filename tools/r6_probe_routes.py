"""Which of the generator's 3x3 layers differ between the FusedAdamW and the torch.optim.AdamW route of tests/test_step_gpu.py::
test_torch_adamw_drives_the_same_step, and are two runs of one route identical?  (round 6 probe)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import mtdgan_oracle as orc
from mtd_gan_amd import engine, kernels as K
from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
from mtd_gan_amd.module.weight_methods import WeightMethods
from mtd_gan_amd.optimizers import FusedAdamW
dev = torch.device("cuda")
x, y = orc.synthetic_ldct(4, seed=5)


def run(kind):
    torch.manual_seed(3)
    m = MTD_GAN_Method().cuda().train()
    mk = (lambda ps: FusedAdamW(ps, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)) if kind == "fused" else \
         (lambda ps: torch.optim.AdamW(ps, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4))
    oD, oG = mk(m.Discriminator.parameters()), mk(m.Generator.parameters())
    wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
    torch.manual_seed(100)
    random.seed(200)
    grads = {}
    hook = oG.register_step_pre_hook(lambda opt, a, k: grads.update({n: p.grad.detach().clone() for n, p in m.Generator.named_parameters()}))
    engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, dev, 0, 0, 4, wm)
    hook.remove()
    return {k: v.detach().clone() for k, v in m.state_dict().items()}, grads


print("BLOCK_FWD_WINO", K.BLOCK_FWD_WINO, "BLOCK_BWD_WINO", K.BLOCK_BWD_WINO, "WINO_C32_BWD", K.WINO_C32_BWD)
(f1, gf1), (f2, gf2), (t1, gt1) = run("fused"), run("fused"), run("torch")
print("fused == fused:", all(torch.equal(f1[k], f2[k]) for k in f1), " gradients equal:", all(torch.equal(gf1[k], gf2[k]) for k in gf1))
ndiff = sum(0 if torch.equal(gf1[k], gt1[k]) else 1 for k in gf1)
print("generator gradient tensors that differ between the routes:", ndiff, "of", len(gf1))
worst = []
for k in gf1:
    if not torch.equal(gf1[k], gt1[k]):
        d = (gf1[k].double() - gt1[k].double()).abs().max().item()
        worst.append((d / (gt1[k].abs().max().item() + 1e-30), k, gt1[k].abs().max().item()))
for r, k, mx in sorted(worst, reverse=True)[:8]:
    print(f"  {k:40s} max |dg| / max |g| = {r:.2e}   max |g| = {mx:.2e}")
for k in f1:
    if k.startswith("Generator.") and tuple(f1[k].shape) == (32, 32, 3, 3):
        d = (f1[k].double() - t1[k].double()).abs()
        fr = (d > 2e-7).double().mean().item()
        if fr > 0.003:
            print(f"  {k:40s} fraction of weights apart by > 2e-7: {fr:.4f}  max {d.max().item():.2e}")
