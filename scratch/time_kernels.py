"""Scratch: time the MLP kernels alone on one 32768x128 chunk (not part of the product)."""
import sys, os, time
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
idn = [0, 2, 4, 6, 8, 10, 12]
nw = [T(st[f'non_rigid_mlp.module.block_mlps.{i}.weight']) for i in idn]
nb = [T(st[f'non_rigid_mlp.module.block_mlps.{i}.bias']) for i in idn]
P = 32768 * 128
x = (torch.rand(P, 3, device=dev) * 2 - 1)
hw = torch.ones(6, device=dev)
cond = torch.zeros(69, device=dev)
for mode in sys.argv[1:] or ['f32', 'f16x3']:
    cp = ops.canonical_pack(cw, cb, mode)
    npk = ops.nonrigid_pack(nw, nb, cond, mode)
    for name, fn in (('canonical', lambda: ops.canonical(x, cp, mode)), ('nonrigid', lambda: ops.nonrigid(x, hw, npk, mode))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        mac = 492032 if name == 'canonical' else 100352
        print(f'{mode:6s} {name:10s} {ms:8.3f} ms  {2*mac*P/ms/1e9:8.1f} TFLOP/s-equiv')
