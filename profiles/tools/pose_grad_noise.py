"""Scratch: how much the pose-decoder gradient of the gradient-golden problem moves with last-bit changes of the forward."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd.config import cfg
from humannerf_amd import network as N
from humannerf_amd.seeded import default_shapes, seeded_state
from tests.test_grad_oracle import grad_frame, reference_loss
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')
meta = json.load(open(os.path.join(gd, 'meta.json')))['grad_s64']
g = np.load(os.path.join(gd, 'grad_s64.npz'))
fr = grad_frame(meta)
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = meta['N_samples'], 0.0, False
fused = N.BodyPoseRefiner.rvec
def run(mode, route, bump=0.0):
    cfg.amd.train_mlp_mode = cfg.amd.train_dw_mode = cfg.amd.train_chain_mode = mode
    cfg.amd.train_operands = 'f32'
    N.BodyPoseRefiner.rvec = fused if route == 'kernel' else (lambda self, x: self.block_mlps(x).view(-1, 3))
    net = N.Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).train()
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
    if bump:
        data['dst_posevec'] = data['dst_posevec'] * (1.0 + bump)
    out = net(**data, iter_val=meta['iter_val'])
    reference_loss(out, torch.from_numpy(g['loss_weights']).to(dev)).backward()
    return {k: p.grad.double().cpu().numpy().ravel() for k, p in net.named_parameters()}
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for mode in ('f32', 'f16x3'):
    A = run(mode, 'torch'); A2 = run(mode, 'torch'); B = run(mode, 'kernel'); C = run(mode, 'torch', bump=1.2e-7)
    for k in ('pose_decoder.block_mlps.0.bias', 'pose_decoder.block_mlps.8.weight', 'non_rigid_mlp.module.block_mlps.0.weight', 'cnl_mlp.module.pts_linears.0.bias', 'cnl_mlp.module.pts_linears.7.weight' if 'cnl_mlp.module.pts_linears.7.weight' in A else 'cnl_mlp.module.pts_linears.8.weight'):
        print(mode, k, 'torch rerun %.1e | kernel vs torch %.1e | torch with posevec*(1+1.2e-7) vs torch %.1e' % (rel(A2[k], A[k]), rel(B[k], A[k]), rel(C[k], A[k])))
