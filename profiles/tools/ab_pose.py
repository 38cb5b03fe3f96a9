"""Scratch: A/B of a switch inside the training step (same process, same box); currently the decoder's 1x1x1 layer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import scene, network as N
from humannerf_amd.config import cfg
from humannerf_amd.train import Trainer
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
idx = torch.from_numpy(np.concatenate([((np.arange(96 + 48 * k, 128 + 48 * k)[:, None]) * 512 + np.arange(80 + 56 * k, 112 + 56 * k)[None]).reshape(-1) for k in range(6)])).to(dev)
tb = dict(data); tb['rays'] = data['rays'][:, idx].contiguous(); tb['near'] = data['near'][idx].contiguous(); tb['far'] = data['far'][idx].contiguous()
tb['target_rgbs'] = torch.rand(6144, 3, device=dev)
cfg.perturb, cfg.N_samples, cfg.train.lossweights.lpips = 1.0, 128, 0.0
fused = N.BodyPoseRefiner.rvec
torch_route = lambda self, x: self.block_mlps(x).view(-1, 3)
net = N.Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).train()
tr = Trainer(net)
for _ in range(5):
    tr.train_step(tb)
for rnd in range(3):
    for name, f in (('point-conv on', True), ('point-conv off', False)):
        N._POINT_CONV = f
        for _ in range(3):
            tr.train_step(tb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            tr.train_step(tb)
        torch.cuda.synchronize()
        print(rnd, name, '%.3f ms' % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
