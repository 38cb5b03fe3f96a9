"""Scratch: K1 outputs of the library named by HNRF_LIB_PATH on fixed random inputs (samples well inside, on the border and far
outside the weight volumes; 24 and 7 bones; ragged sample count) -> a checksum file, to compare two builds bit for bit.
    HNRF_LIB_PATH=a.so python profiles/tools/k1_bits.py /tmp/a.pt; HNRF_LIB_PATH=b.so python ... /tmp/b.pt; cmp"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from humannerf_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(5)
out = {}
for B, R, S in ((24, 3001, 128), (24, 17, 7), (7, 513, 64)):
    G = 32
    rays_o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    near = torch.full((R,), 0.1) + torch.rand(R, generator=g) * 0.1
    far = near + 2.5 + torch.rand(R, generator=g)
    t_rand = torch.rand(R, S, generator=g)
    A = torch.randn(B, 3, 3, generator=g) * 0.3 + torch.eye(3)
    T = torch.randn(B, 3, generator=g) * 0.3
    vol = torch.softmax(torch.randn(B + 1, G, G, G, generator=g) * 2, dim=0).contiguous()
    bmin = torch.tensor([-0.9, -1.1, -0.7]); bscale = torch.tensor([2 / 1.8, 2 / 2.2, 2 / 1.4])
    args = [t.to(dev).contiguous() for t in (rays_o, rays_d, near, far, t_rand, A, T, vol, bmin, bscale)]
    for tr in (args[4], None):
        z, xs, m, w = ops.sample_warp(args[0], args[1], args[2], args[3], tr, args[5], args[6], args[7], args[8], args[9], S, want_bmw=True)
        z2, xs2, m2, _ = ops.sample_warp(args[0], args[1], args[2], args[3], tr, args[5], args[6], args[7], args[8], args[9], S)
        print('   lean == diagnostic form:', bool(torch.equal(z, z2)), bool(torch.equal(xs, xs2)), bool(torch.equal(m, m2)), float((xs - xs2).abs().max()))
        out['%d_%d_%d_%s' % (B, R, S, tr is not None)] = [t.cpu() for t in (z, xs, m, w)]
        print(B, R, S, 'fg>0: %.3f' % float((m > 0).float().mean()), 'sum', float(m.double().sum()))
torch.save(out, sys.argv[1])
