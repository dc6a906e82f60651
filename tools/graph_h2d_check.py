import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtd_gan_amd import kernels as K
dev = torch.device("cuda")
slot = K.HostScalars(dev, 4, torch.float32)
out = torch.zeros(4, device=dev)
slot.set([1, 2, 3, 4])
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    out.copy_(slot.upload())
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    d = slot.upload()
    out.copy_(d * 2)
for vals in ([5, 6, 7, 8], [9, 10, 11, 12]):
    slot.set(vals)
    g.replay()
    torch.cuda.synchronize()
    print("host", slot.host.tolist(), "dev", slot.dev.tolist(), "out", out.tolist())
