"""Scratch: wall-time split of one training iteration, per storage mode of the weight-gradient operands.
    python profiles/tools/time_train.py [f16|f32] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.train import Trainer, image_loss, update_lr
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
cfg.amd.train_operands = sys.argv[1] if len(sys.argv) > 1 else 'f16'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg.amd.train_check_every = int(os.environ.get('HNRF_CHECK_EVERY', 200))
state = seeded_state(default_shapes(), 0)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).train()
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
_idx = []
for k in range(6):
    y0, x0 = 96 + 48 * k, 80 + 56 * k
    yy, xx = np.meshgrid(np.arange(y0, y0 + 32), np.arange(x0, x0 + 32), indexing='ij')
    _idx.append((yy * 512 + xx).reshape(-1))
idx = torch.from_numpy(np.concatenate(_idx)).to(dev)
tb = dict(data); tb['rays'] = data['rays'][:, idx].contiguous(); tb['near'] = data['near'][idx].contiguous(); tb['far'] = data['far'][idx].contiguous()
tgt = torch.rand(6144, 3, device=dev)
cfg.perturb, cfg.N_samples, cfg.train.lossweights.lpips = 1.0, 128, 0.0
tr = Trainer(net)
START = int(os.environ.get('HNRF_START_ITER', 60000))         # steady state: past kick_in_iter / full_band_iter (1 = the cheaper early step)
tr.iter = START
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(iters):
    t0 = sync(); tr.optimizer.zero_grad(set_to_none=True)
    out = net(**tb, iter_val=float(START + it)); t1 = sync()
    loss, _ = image_loss(out['rgb'][None, None], tgt[None, None]); t2 = sync()
    loss.backward(); t3 = sync()
    tr.optimizer.step(); t4 = sync()
    update_lr(tr.optimizer, it + 1); t5 = sync()
    print('operands %s iter %d: fwd %.2f loss %.2f bwd %.2f opt %.2f ms | total %.2f' % (cfg.amd.train_operands, it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t0)*1e3))
tb['target_rgbs'] = tgt
t0 = sync()
for it in range(10):
    tr.train_step(tb)
t1 = sync()
print('operands %s: %.2f ms per train_step (10 back to back)' % (cfg.amd.train_operands, (t1 - t0) * 100))
