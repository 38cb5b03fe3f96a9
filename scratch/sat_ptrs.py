import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import ops, _lib
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = {k: torch.from_numpy(v).to(dev) for k, v in seeded_state(default_shapes(), 0).items()}
P = 4096
xyz = (torch.rand(P, 3, device=dev) - 0.5)
ws = [st['cnl_mlp.module.pts_linears.%d.weight' % i] for i in range(0, 16, 2)] + [st['cnl_mlp.module.output_linear.0.weight']]
bs = [st['cnl_mlp.module.pts_linears.%d.bias' % i] for i in range(0, 16, 2)] + [st['cnl_mlp.module.output_linear.0.bias']]
lib = _lib.load()
print('bytes', lib.hnrf_canonical_packed_bytes(1), lib.hnrf_nonrigid_packed_bytes(1), 'status', lib.hnrf_canonical_status_offset(1), lib.hnrf_nonrigid_status_offset(1))
for s in (1.0, 1.0):
    w2 = [w.clone() for w in ws]
    packed = ops.canonical_pack(w2, bs, 'f16x3')
    print('canonical packed %x numel*4 %d status addr %x' % (packed.data_ptr(), packed.numel() * 4, packed.data_ptr() + lib.hnrf_canonical_status_offset(1)))
    raw = ops.canonical(xyz, packed, 'f16x3'); torch.cuda.synchronize()
nw = [st['non_rigid_mlp.module.block_mlps.%d.weight' % i] for i in range(0, 14, 2)]
nb = [st['non_rigid_mlp.module.block_mlps.%d.bias' % i] for i in range(0, 14, 2)]
hann = torch.ones(6, device=dev); cond = torch.randn(69, device=dev) * 0.1
for s in (1.0, 1.0):
    w2 = [w.clone() for w in nw]
    packed = ops.nonrigid_pack(w2, nb, cond, 'f16x3')
    print('nonrigid packed %x numel*4 %d status addr %x' % (packed.data_ptr(), packed.numel() * 4, packed.data_ptr() + lib.hnrf_nonrigid_status_offset(1)))
    out, _ = ops.nonrigid(xyz, hann, packed, 'f16x3'); torch.cuda.synchronize()
print(torch.cuda.memory_snapshot()[:3] if False else '')
