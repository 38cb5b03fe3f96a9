"""Scratch: full 11-key output path (cfg.amd.diagnostics = True, the reference's return signature) vs lean path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
state = seeded_state(default_shapes(), 0)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); net = net.to(dev).eval()
fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
cfg.perturb, cfg.N_samples = 0., 128
for diag in (False, True):
    cfg.amd.diagnostics = diag
    with torch.no_grad():
        out = net(**data, iter_val=1e7); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): out = net(**data, iter_val=1e7)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('diagnostics=%s: %d keys, %.1f ms/frame = %.2f M rays/s, peak mem %.1f GB' % (diag, len(out), dt * 1e3, 262144 / dt / 1e6, torch.cuda.max_memory_allocated() / 2**30))
    del out
