"""Scratch: cycle stamps of the two-group non-rigid kernel (diagnostic build, HNRF_LIB_PATH=profiles/tools/libhnrf_stamp.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import ops, _lib
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = seeded_state({k: v for k, v in default_shapes().items() if k.startswith('non_rigid_mlp')}, 0)
T = lambda a: torch.from_numpy(a).to(dev)
idx = [0, 2, 4, 6, 8, 10, 12]
w = [T(st[f'non_rigid_mlp.module.block_mlps.{i}.weight']) for i in idx]
b = [T(st[f'non_rigid_mlp.module.block_mlps.{i}.bias']) for i in idx]
nbytes = _lib.load().hnrf_nonrigid_packed_bytes(1)
buf = torch.zeros(nbytes // 4 + 4096 * 8 + 64, device=dev)
cond = torch.randn(69, device=dev) * 0.1
pk = ops.nonrigid_pack(w, b, cond, 'f16x3', out=buf)
assert pk.data_ptr() == buf.data_ptr()
P = 32768 * 128
x = torch.rand(P, 3, device=dev) * 2 - 1
hann = torch.ones(6, device=dev)
for _ in range(2):
    ops.nonrigid(x, hann, pk, 'f16x3+noguard')
torch.cuda.synchronize()
dbg = buf.view(torch.int64)[nbytes // 8: nbytes // 8 + 4096 * 4].cpu().numpy().reshape(4096, 4)
k, o = dbg[:, 0].astype(np.float64), dbg[:, 1].astype(np.float64)
print('K2 x2, per workgroup-wave (256 samples): k-loop cycles median %.0f (ideal 1152 MFMAs x 32 = 36864), outside median %.0f' % (np.median(k), np.median(o)))
print('per tile (25): k %.0f, other %.0f' % (np.median(k) / 25, np.median(o) / 25))
