"""Scratch: dW kernel vs torch, and timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from humannerf_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
def bench(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for P in (1000, 786432):
    for n_out, n_in, ldx in ((256, 256, 256), (256, 63, 63), (128, 128, 128), (128, 36, 36), (256, 128, 128), (128, 256, 256), (4, 256, 256), (3, 128, 128)):
        dZ = torch.randn(P, n_out, device=dev) * (torch.rand(P, n_out, device=dev) > 0.5)
        X = torch.relu(torch.randn(P, ldx, device=dev))
        ref_w = (dZ.double().t() @ X.double()); ref_b = dZ.double().sum(0)
        w, b = ops.mlp_dw(dZ, X)
        if n_out >= 128 and n_in >= 128:
            sc = 10.0 ** np.random.uniform(-9, 2)
            am = (dZ * sc).abs().amax().reshape(1)
            w16, b16 = ops.mlp_dw(dZ * sc, X, mode='f16x3', dz_amax=am)
            e16 = float((w16.double() / sc - ref_w).abs().max() / ref_w.abs().max()); eb16 = float((b16.double() / sc - ref_b).abs().max() / ref_b.abs().max())
            extra = f'  f16x3 err {e16:.2e} db {eb16:.2e}'
            if P > 1000: extra += ' %.3f ms' % bench(lambda: ops.mlp_dw(dZ, X, mode='f16x3', dz_amax=am))
        else: extra = ''
        ew = float((w.double() - ref_w).abs().max() / ref_w.abs().max()); eb = float((b.double() - ref_b).abs().max() / ref_b.abs().max())
        tw = (dZ.t() @ X).double(); et = float((tw - ref_w).abs().max() / ref_w.abs().max())
        line = f'P={P} {n_out}x{n_in}: rel err dW {ew:.2e} (torch {et:.2e}) db {eb:.2e}'
        if P > 1000:
            line += '  hnrf %.3f ms  torch %.3f ms' % (bench(lambda: ops.mlp_dw(dZ, X)), bench(lambda: (dZ.t() @ X, dZ.sum(0))))
        print(line + extra)
# skip-layer style: write into a column block of a wider matrix
P = 5000
dZ = torch.randn(P, 256, device=dev); pe = torch.randn(P, 63, device=dev); hcat = torch.randn(P, 256, device=dev)
full = torch.zeros(256, 319, device=dev)
ops.mlp_dw(dZ, pe, full[:, :63], want_db=False); ops.mlp_dw(dZ, hcat, full[:, 63:])
ref = dZ.t() @ torch.cat([pe, hcat], 1)
print('skip block err', float((full - ref).abs().max() / ref.abs().max()))
