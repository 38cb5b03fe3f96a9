"""Scratch: what fraction of samples has an exactly-zero gradient at the canonical MLP output in a training step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import scene, ops
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).train()
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
idx = []
for k in range(6):
    y0, x0 = 96 + 48 * k, 80 + 56 * k
    yy, xx = np.meshgrid(np.arange(y0, y0 + 32), np.arange(x0, x0 + 32), indexing='ij')
    idx.append((yy * 512 + xx).reshape(-1))
idx = torch.from_numpy(np.concatenate(idx)).to(dev)
tb = dict(data); tb['rays'] = data['rays'][:, idx].contiguous(); tb['near'] = data['near'][idx].contiguous(); tb['far'] = data['far'][idx].contiguous()
orig = ops.composite_bwd
def spy(*a, **k):
    d_raw, d_mask = orig(*a, **k)
    z = (d_raw.reshape(-1, 4) == 0).all(1).float().mean().item()
    tiny = (d_raw.reshape(-1, 4).abs().amax(1) < 1e-12 * d_raw.abs().max()).float().mean().item()
    print('rows with d_raw == 0 exactly: %.1f %%; below 1e-12 of the largest: %.1f %%' % (100 * z, 100 * tiny))
    return d_raw, d_mask
ops.composite_bwd = spy
cfg.perturb, cfg.N_samples = 1.0, 128
out = net(**tb, iter_val=10.0)
(out['rgb'] - torch.rand_like(out['rgb'])).pow(2).mean().backward()
