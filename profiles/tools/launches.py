"""Scratch: which LINES of the host code launch the small library kernels of a training step (fills, copies, reductions)?
torch.profiler over a few steady-state iterations, grouped by (operator, shapes, innermost frame inside humannerf_amd/).
    python profiles/tools/launches.py [iters]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from humannerf_amd import scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.train import Trainer
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
tb['target_rgbs'] = torch.rand(6144, 3, device=dev)
cfg.perturb, cfg.N_samples, cfg.train.lossweights.lpips = 1.0, 128, 0.0
tr = Trainer(net)
tr.iter = 60000
for _ in range(5):
    tr.train_step(tb)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(iters):
        tr.train_step(tb)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, 'device_time_total', 0) or getattr(e, 'cuda_time_total', 0)
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    kt = sum(k.duration for k in e.kernels)
    where = next((s for s in (e.stack or []) if 'humannerf_amd/' in s), (e.stack or ['?'])[0] if e.stack else '?')
    where = where.split('humannerf_amd/')[-1][:60]
    agg[(e.name, str(e.input_shapes)[:70], where)][0] += len(e.kernels)
    agg[(e.name, str(e.input_shapes)[:70], where)][1] += kt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('device time of operator-launched kernels: %.3f ms per iteration' % (tot / iters / 1e3))
for (name, shapes, where), (n, t) in rows[:70]:
    print('%8.4f ms %5.1f x  %-34s %-70s %s' % (t / iters / 1e3, n / iters, name[:34], shapes, where))
