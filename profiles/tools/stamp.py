"""Scratch: read the diagnostic cycle stamps of the f16x3 canonical kernel (HNRF_LIB_PATH=profiles/tools/libhnrf_stamp.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import ops, _lib
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = seeded_state({k: v for k, v in default_shapes().items() if k.startswith('cnl_mlp')}, 0)
T = lambda a: torch.from_numpy(a).to(dev)
idx = [0, 2, 4, 6, 8, 10, 12, 14]
cw = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
cb = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
nbytes = _lib.load().hnrf_canonical_packed_bytes(1)
buf = torch.zeros(nbytes // 4 + 4096 * 8 + 64, device=dev)
cp = ops.canonical_pack(cw, cb, 'f16x3', out=buf)
assert cp.data_ptr() == buf.data_ptr()
P = 32768 * 128
x = torch.rand(P, 3, device=dev) * 2 - 1
for _ in range(2):
    ops.canonical(x, cp, 'f16x3')
torch.cuda.synchronize()
dbg = buf.view(torch.int64)[nbytes // 8: nbytes // 8 + 4096 * 4].cpu().numpy().reshape(4096, 4)
k, b = dbg[:, 0].astype(np.float64), dbg[:, 1].astype(np.float64)
print('k-loop cycles per WG-wave: median %.0f  (ideal 2928*32 = 93696)' % np.median(k))
print('outside k-loops:           median %.0f' % np.median(b))
print('per slab step (59): k %.0f, other %.0f' % (np.median(k) / 59, np.median(b) / 59))
if dbg[:, 2].any():
    print('of the other: DMA wait (s_waitcnt vmcnt) median %.0f, barrier median %.0f (per step %.0f / %.0f)' % (
        np.median(dbg[:, 2]), np.median(dbg[:, 3]), np.median(dbg[:, 2]) / 59, np.median(dbg[:, 3]) / 59))
