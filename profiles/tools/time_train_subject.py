"""End-to-end training rate from a prepared subject directory (synthetic subject, 16 frames, rendered at 512x512):
data side = dataset.FrameStream with the device-resident frame cache vs the numpy route.
    python profiles/tools/time_train_subject.py [iters] [lens]
``lens``: 1024x1024 source PNGs with lens distortion and cfg.resize_img_scale = 0.5 (the ZJU-387 / wild setting)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
from humannerf_amd.train import Trainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# "lens": what every 387 / wild yaml configures -- 1024x1024 PNGs, `distortions` on every camera, resize_img_scale 0.5
lens = len(sys.argv) > 2 and sys.argv[2] == 'lens'
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=16, size=1024 if lens else 512, binary_mask=True,
                              distortions=scene.ZJU_LIKE_DISTORTION if lens else None)
cfg.resize_img_scale = 0.5 if lens else 1.0

dev = torch.device('cuda:0')
cfg.train.lossweights.lpips, cfg.N_samples = 0.0, 128
subj = dataset.Subject(d)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()})
net = net.to(dev)
for cache in (True, False):
    tr = Trainer(net)
    tr.iter = int(os.environ.get('HNRF_START_ITER', 60000))   # steady state: past kick_in_iter / full_band_iter (1 = the cheaper early step)
    stream = dataset.FrameStream(subj, device=dev, device_cache=cache, workers=int(os.environ.get("W", 3)), prefetch=int(os.environ.get("PF", 4)))
    cfg.perturb = cfg.train.perturb
    n = iters if cache else max(8, iters // 5)
    for i in range(20 if cache else 4):                      # warm-up: kernels, frame cache (16 frames)
        tr.train_step(next(stream))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rays = 0
    for i in range(n):
        b = next(stream)
        rays += b['rays'].shape[1]
        tr.train_step(b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stream.close()
    print('device_cache=%s: %.2f ms per iteration (%.1f it/s), %.0f rays per item, cache %.1f MB' % (
        cache, dt / n * 1e3, n / dt, rays / n, (stream.cache.bytes / 2 ** 20) if stream.cache else 0.0), flush=True)
