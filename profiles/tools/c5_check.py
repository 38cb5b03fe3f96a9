"""BASELINE config 5 (1024 x 1024 x 256 samples, 268 M samples per frame) end to end: time, memory, and
consistency of a strided subset of its rays rendered alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import scene, ops
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).eval()
fr = scene.synthetic_frame(H=1024, W=1024, focal_at_512=1700.0, camera_only=True)
rays = ops.gen_rays(fr['K'], fr['E'], fr['cnl_bbox_min_xyz'], fr['cnl_bbox_max_xyz'], 1024, 1024)
keys = ['dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
data.update(rays=rays['rays'], near=rays['near'], far=rays['far'])
cfg.perturb, cfg.N_samples, cfg.amd.diagnostics = 0., 256, False
with torch.no_grad():
    net(**data, iter_val=1e7); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = net(**data, iter_val=1e7); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    N = data['rays'].shape[1]
    print('C5: %d rays x 256 samples in %.1f ms = %.2f M rays/s (%.2f M rays/s at the 128-sample cost); peak memory %.2f GB'
          % (N, dt * 1e3, N / dt / 1e6, 2 * N / dt / 1e6, torch.cuda.max_memory_allocated() / 2**30))
    sel = torch.arange(0, N, 4099, device=dev)
    sub = dict(data); sub['rays'] = data['rays'][:, sel].contiguous(); sub['near'] = data['near'][sel].contiguous(); sub['far'] = data['far'][sel].contiguous()
    o2 = net(**sub, iter_val=1e7)
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(out[k][sel], o2[k]), k
    print('subset of %d rays rendered alone: bit-identical' % sel.numel())
