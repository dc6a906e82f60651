import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch
import mtdgan_oracle as orc
import mtd_gan_amd
from mtd_gan_amd.arch.Ours.networks import ResFFT_Generator
g = orc.seeded_fill(orc.g_param_shapes(), seed=7)
G = ResFFT_Generator(1, 32, 10, 3, 1)
G.load_state_dict(g); G.cuda().eval()
for S, B in ((128, 2), (256, 1), (512, 1)):
    x, y = orc.synthetic_ldct(B, seed=5, size=S)
    t0 = time.time()
    ref = orc.generator_forward(g, x)
    tc = time.time() - t0
    with torch.no_grad():
        out = G(x.cuda())
        torch.cuda.synchronize()
        t0 = time.time()
        out = G(x.cuda()); torch.cuda.synchronize()
        tg = time.time() - t0
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"S={S} B={B} rel err {err:.2e}  psnr diff {abs(orc.psnr(out.cpu().clip(0,1), y) - orc.psnr(ref.clip(0,1), y)).item():.4f} dB   cpu {tc:.2f}s gpu {tg*1e3:.1f} ms")
