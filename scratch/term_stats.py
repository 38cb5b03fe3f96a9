"""Scratch: how many samples does early ray termination save on the bench frame, on top of culling?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import scene, ops
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).eval()
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
cfg.perturb, cfg.N_samples, cfg.amd.diagnostics = 0., 128, True
with torch.no_grad():
    out = net(**data, iter_val=1e7)
w = out['weights_on_rays']; bmw = out['backward_motion_weights'].sum(-1)
alpha_s = None
T = torch.cumprod(torch.cat([torch.ones_like(w[:, :1]), 1 - (w / (torch.cumprod(torch.cat([torch.ones_like(w[:, :1]), torch.ones_like(w[:, :-1])], 1), 1)))[:, :-1]], 1), 1) if False else None
# transmittance in front of each sample from the weights: T_i = 1 - sum_{j<i} w_j (up to the 1e-10 terms)
Tfront = 1.0 - torch.cumsum(w, 1) + w
P = w.numel()
culled = (bmw < 1e-9)
for eps in (1e-2, 1e-3, 1e-4):
    # slab granularity: a ray stops at the first slab boundary after T < eps
    dead = torch.zeros_like(culled)
    for s0 in range(32, 128, 32):
        dead[:, s0:] |= (Tfront[:, s0:s0 + 1] < eps)
    print('term_eps %g: culled %.1f %%, dead-by-termination %.1f %%, evaluated with both %.1f %%' % (
        eps, 100 * culled.float().mean().item(), 100 * dead.float().mean().item(), 100 * (~culled & ~dead).float().mean().item()))
print('rays with final alpha > 0.999: %.1f %%' % (100 * (out['alpha'] > 0.999).float().mean().item()))
