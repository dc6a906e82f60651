"""The price of the Winograd conv's split of K on the step's mid-size shapes (round 6, verdict item 4: "a single-layer prototype table").
Times one F(2x4, 3x3) launch + its slab-sum launch per shape with the split the plan picks and with a forced one (the lab library reads
MTD_WINO_SPLITK once per process, so every setting is a child process):   MTD_LAB=1 python tools/splitk_price.py
Columns: us per conv (kernel + splitk_epilogue where S > 1; hot operands, 30 repetitions) for S = plan | 1 | 2 | 4 | 8."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 512, 512, 8), (64, 256, 256, 8), (64, 128, 128, 16), (64, 512, 128, 16), (64, 1024, 256, 8), (64, 512, 256, 8), (64, 256, 128, 16),
          (64, 256, 512, 8), (32, 512, 512, 8), (32, 256, 256, 16)]      # (images, C, N, map side)


def child():
    sys.path.insert(0, ROOT)
    import torch
    from mtd_gan_amd import _lib, kernels as K
    assert _lib.lib().mtd_lab_build(), "needs the lab library (MTD_LAB_BUILD=1 python mtd-gan_amd/_build.py; MTD_LAB=1)"
    for (B, Ci, Co, H) in SHAPES:
        x = torch.randn(B, H, H, Ci, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * (9 * Ci) ** -0.5
        out = torch.empty(B, H, H, Co, device="cuda")
        geom = K.geom_fwd(B, H, H, 3, 1, 1)
        for _ in range(3):
            K.conv(x, w, geom, Co, Ci, Ci * 9, 9, out, act=K.ACT_LRELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            K.conv(x, w, geom, Co, Ci, Ci * 9, 9, out, act=K.ACT_LRELU)
        e1.record()
        torch.cuda.synchronize()
        a = K._conv_args(x, w, geom, Co, Ci, Ci * 9, 9, out, pack=False, count=False)
        a.w_st = K.winograd_patch_w(geom, Co, Ci)
        ws = _lib.lib().mtd_conv_winograd_ws_bytes(_lib.C.byref(a))
        S = ws // (B * H * H * Co * 4) if ws else 1
        print(f"{B * H * H} {Co} {Ci} {S} {1e3 * e0.elapsed_time(e1) / 30:.1f}", flush=True)


def main():
    table = {}
    for sk in ("plan", "1", "2", "4", "8"):
        env = dict(os.environ, MTD_LAB="1", MTD_SPLITK_PRICE_CHILD="1")
        env.pop("MTD_WINO_SPLITK", None)
        if sk != "plan":
            env["MTD_WINO_SPLITK"] = sk
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit(1)
        for ln in r.stdout.splitlines():
            M, N, Cc, S, us = ln.split()
            table.setdefault((int(M), int(N), int(Cc)), {})[sk] = (int(S), float(us))
    print("wino_conv_kernel<2, false, 6> + splitk_epilogue, us per layer (S = the split of K the launch used)")
    print(f"{'M':>7} {'N':>5} {'C':>5} | " + " | ".join(f"{k:>14}" for k in ("plan", "S=1", "S=2", "S=4", "S=8")))
    for (M, N, Cc), row in table.items():
        print(f"{M:7d} {N:5d} {Cc:5d} | " + " | ".join(f"S={row[k][0]:<2d} {row[k][1]:8.1f} us" for k in ("plan", "1", "2", "4", "8")))


if __name__ == "__main__":
    child() if os.environ.get("MTD_SPLITK_PRICE_CHILD") == "1" else main()
