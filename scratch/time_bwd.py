"""Scratch: time the backward chain kernels alone (786432 samples)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import ops
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = seeded_state({k: v for k, v in default_shapes().items() if 'mlp' in k and 'decoder' not in k}, 0)
T = lambda a: torch.from_numpy(a).to(dev)
idx = [0, 2, 4, 6, 8, 10, 12, 14]
cw = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
cb = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
P = 6144 * 128
x = (torch.rand(P, 3, device=dev) * 2 - 1)
cp = ops.canonical_pack(cw, cb, 'f32')
raw, pe, acts, bits = ops.canonical_train(x, cp)
g = torch.randn(P, 4, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('canonical fwd (infer) %.3f ms' % t(lambda: ops.canonical(x, cp, 'f32')))
print('canonical fwd (train) %.3f ms' % t(lambda: ops.canonical_train(x, cp)))
print('canonical bwd chain   %.3f ms' % t(lambda: ops.canonical_bwd(x, g, bits, cw)))
cp16 = ops.canonical_pack(cw, cb, 'f16x3')
print('canonical fwd f16x3 (infer) %.3f ms' % t(lambda: ops.canonical(x, cp16, 'f16x3')))
print('canonical fwd f16x3 (train) %.3f ms' % t(lambda: ops.canonical_train(x, cp16, 'f16x3')))
print('canonical bwd chain f16x3 %.3f ms' % t(lambda: ops.canonical_bwd(x, g, bits, cw, 'f16x3')))
dz32, dx32, am32 = ops.canonical_bwd(x, g * 1e-7, bits, cw, 'f32')
dz16, dx16, am16 = ops.canonical_bwd(x, g * 1e-7, bits, cw, 'f16x3')
print('f16x3 vs f32 chain at |d_raw| ~ 1e-7: dZ rel diff %.2e, d_xyz rel diff %.2e' % (float((dz16 - dz32).abs().max() / dz32.abs().max()), float((dx16 - dx32).abs().max() / dx32.abs().max())))
