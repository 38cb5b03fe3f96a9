"""Scratch: per-step wall times of 260 training steps (looking for periodic stalls)."""
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
net = N.Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).train()
tr = Trainer(net)
import gc
if os.environ.get("NOGC"): gc.freeze(); gc.disable()
ts = []
for i in range(260):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(tb)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print('median %.2f ms; steps above 1.3x median:' % np.median(ts[5:]), [(i, round(float(t), 1)) for i, t in enumerate(ts) if i >= 5 and t > 1.3 * np.median(ts[5:])])
print('reserved MB', torch.cuda.memory_reserved() / 2**20, 'allocated', torch.cuda.memory_allocated() / 2**20)
