"""Scratch: frame loop (render_frames over camera-only frames) against the pure render rate of the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import scene
import importlib
render = importlib.import_module("humannerf_amd." + os.environ.get("RENDER_MOD", "render"))
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()}); net = net.to(dev).eval()
cams = [scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, pose_seed=i % 3, camera_only=True) for i in range(12)]
full = scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, pose_seed=0)
n_rays = full['rays'].shape[1]
keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
data = {k: torch.from_numpy(np.ascontiguousarray(full[k])).to(dev) for k in keys}
cfg.perturb = 0.
with torch.no_grad():
    for _ in range(2): net(**data, iter_val=1e7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): net(**data, iter_val=1e7)
    torch.cuda.synchronize(); pure = (time.perf_counter() - t0) / 6
render.render_frames(net, cams[:2], device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
render.render_frames(net, cams, device=dev)
torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / len(cams)
print('pure render of the %d hit rays: %.2f ms; frame loop: %.2f ms per frame (%.2f fps); overhead %.2f ms' % (n_rays, pure * 1e3, loop * 1e3, 1 / loop, (loop - pure) * 1e3))
