"""Scratch: element-wise gradient error of the training kernels vs the fp64 oracle, per operand mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
from oracle import oracle
dev = torch.device('cuda:0')
S = 64
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 61
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, ray_stride=stride)
R = fr['rays'].shape[1]
rs = np.random.RandomState(21)
target = rs.rand(R, 3).astype(np.float32)
t_rand = rs.rand(R, S).astype(np.float32)
sp = seeded_state(default_shapes(), 0)
KEYS = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
params = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sp.items()}
out = oracle.render(params, fr, iter_val=30000.0, N_samples=S, t_rand=t_rand, dtype=torch.float64)
(0.2 * torch.mean((out['rgb'] - torch.from_numpy(target).double()) ** 2)).backward()
ref = {k: v.grad.numpy() for k, v in params.items()}
cfg.N_samples, cfg.perturb = S, 1.0
for arith, operands in (('f32', 'f32'), ('f16x3', 'f32'), ('f16x3', 'f16')):
    cfg.amd.train_mlp_mode = cfg.amd.train_chain_mode = cfg.amd.train_dw_mode = arith
    cfg.amd.train_operands = operands
    net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in sp.items()}); net = net.to(dev).train()
    batch = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in KEYS}
    o = net(**batch, iter_val=30000.0, t_rand=torch.from_numpy(t_rand).to(dev))
    (0.2 * torch.mean((o['rgb'] - torch.from_numpy(target).to(dev)) ** 2)).backward()
    rows = []
    for k, p in net.named_parameters():
        g, r = p.grad.double().cpu().numpy(), ref[k]
        e = np.abs(g - r).max() / np.abs(r).max()
        flips = int(((np.sign(g) != np.sign(r)) & (np.abs(r) > 1e-2 * np.abs(r).max())).sum())
        rows.append((e, k, flips))
    rows.sort(reverse=True)
    print(arith, operands, 'rays', R, 'worst max|dg|/max|g|:', [(round(e, 6), k.split('.')[0] + '.' + '.'.join(k.split('.')[-2:]), f) for e, k, f in rows[:6]])
