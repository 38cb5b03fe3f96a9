"""Scratch: host numpy ray generation vs hnrf_gen_rays at 512x512 and 1024x1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from humannerf_amd import ops, scene
dev = torch.device('cuda:0')
J = scene.TPOSE_JOINTS
mn, mx = (J.min(0) - 0.3).astype(np.float32), (J.max(0) + 0.3).astype(np.float32)
for H in (512, 1024):
    K, E = scene.tpose_camera(np.array([H, H], dtype=np.float32), 6.0, 1250.0 * H / 512.0)
    t0 = time.perf_counter()
    ro, rd = scene.get_rays_from_KRT(H, H, K, E[:3, :3], E[:3, 3])
    ro, rd = ro.reshape(-1, 3).astype(np.float32), rd.reshape(-1, 3).astype(np.float32).copy()
    near, far, hit = scene.rays_intersect_3d_bbox(np.stack([mn, mx]), ro, rd)
    rays = torch.from_numpy(np.stack([ro[hit], rd[hit], rd[hit]])).to(dev); torch.cuda.synchronize()
    t_np = time.perf_counter() - t0
    ops.gen_rays(K, E, mn, mx, H, H); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g = ops.gen_rays(K, E, mn, mx, H, H)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 10
    print('%dx%d: numpy + upload %.1f ms, device %.3f ms (%d of %d rays kept)' % (H, H, t_np * 1e3, t_gpu * 1e3, int(hit.sum()), H * H))
