import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from humannerf_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(1)
R, S, B, G = 6144, 128, 24, 32
rays_o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
near = torch.full((R,), 0.1); far = near + 2.5
A = torch.randn(B, 3, 3, generator=g) * 0.1 + torch.eye(3); T = torch.randn(B, 3, generator=g) * 0.1
vol = torch.softmax(torch.randn(B + 1, G, G, G, generator=g) * 2, dim=0).contiguous()
bmin = torch.tensor([-0.9, -1.1, -0.7]); bscale = torch.tensor([2 / 1.8, 2 / 2.2, 2 / 1.4])
a = [t.to(dev).contiguous() for t in (rays_o, rays_d, near, far, A, T, vol, bmin, bscale)]
z, xs, m, _ = ops.sample_warp(a[0], a[1], a[2], a[3], None, a[4], a[5], a[6], a[7], a[8], S)
gx = torch.randn(R, S, 3, generator=g).to(dev); gm = torch.randn(R, S, generator=g).to(dev)
for _ in range(3): out = ops.sample_warp_bwd(a[0], a[1], z, a[4], a[5], a[6], a[7], a[8], xs, m, gx, gm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): out = ops.sample_warp_bwd(a[0], a[1], z, a[4], a[5], a[6], a[7], a[8], xs, m, gx, gm)
e1.record(); torch.cuda.synchronize()
print('K1 bwd %s threads: %.4f ms; checksums %.6e %.6e %.6e' % (os.environ.get('HNRF_K1B_THREADS', '256'), e0.elapsed_time(e1) / 20, float(out[0].double().sum()), float(out[1].double().abs().sum()), float(out[2].double().abs().sum())))
