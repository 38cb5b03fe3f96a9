"""f16-range guard probe: canonical / non-rigid inference kernels with one hidden layer scaled up; prints the status words."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import ops
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = {k: torch.from_numpy(v).to(dev) for k, v in seeded_state(default_shapes(), 0).items()}
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 3e5
P = 4096
xyz = (torch.rand(P, 3, device=dev) - 0.5)
ws = [st['cnl_mlp.module.pts_linears.%d.weight' % i] for i in range(0, 16, 2)] + [st['cnl_mlp.module.output_linear.0.weight']]
bs = [st['cnl_mlp.module.pts_linears.%d.bias' % i] for i in range(0, 16, 2)] + [st['cnl_mlp.module.output_linear.0.bias']]
for s in (1.0, scale):
    w2 = [w.clone() for w in ws]
    w2[1] = w2[1] * s
    packed = ops.canonical_pack(w2, bs, 'f16x3')
    word = ops.status_word(packed, 'canonical', 'f16x3')
    print('canonical scale', s, 'status before', int(word.item()), 'packed %x end %x status %x' % (packed.data_ptr(), packed.data_ptr() + packed.numel() * 4, word.data_ptr()), flush=True)
    raw = ops.canonical(xyz, packed, 'f16x3')
    torch.cuda.synchronize()
    print('  after', int(word.item()), 'raw finite', bool(torch.isfinite(raw).all()), 'max', float(raw.abs().max()), flush=True)
nw = [st['non_rigid_mlp.module.block_mlps.%d.weight' % i] for i in range(0, 14, 2)]
nb = [st['non_rigid_mlp.module.block_mlps.%d.bias' % i] for i in range(0, 14, 2)]
hann = torch.ones(6, device=dev)
cond = torch.randn(69, device=dev) * 0.1
for s in (1.0, scale):
    w2 = [w.clone() for w in nw]
    w2[1] = w2[1] * s
    packed = ops.nonrigid_pack(w2, nb, cond, 'f16x3')
    word = ops.status_word(packed, 'nonrigid', 'f16x3')
    print('nonrigid scale', s, 'status before', int(word.item()), 'packed %x end %x status %x xyz %x' % (packed.data_ptr(), packed.data_ptr() + packed.numel() * 4, word.data_ptr(), xyz.data_ptr()), flush=True)
    out, _ = ops.nonrigid(xyz, hann, packed, 'f16x3')
    print('  out %x' % out.data_ptr(), flush=True)
    torch.cuda.synchronize()
    print('  after', int(word.item()), 'finite', bool(torch.isfinite(out).all()), flush=True)
print('DONE', flush=True)
