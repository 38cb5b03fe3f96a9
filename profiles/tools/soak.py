"""Soak: 1500 training iterations from a subject directory (lens setting) + three 60-frame movement renders, watching device
memory, host RSS and the rates over time (leaks, drifting step times, guard false positives).
    python profiles/tools/soak.py"""
import os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, run, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
from humannerf_amd.train import Trainer
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=60, size=1024, binary_mask=True, distortions=scene.ZJU_LIKE_DISTORTION)
cfg.resize_img_scale = 0.5
cfg.train.lossweights.lpips, cfg.N_samples = 0.0, 128
dev = torch.device('cuda:0')
subj = dataset.Subject(d)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()})
net = net.to(dev)
tr = Trainer(net)
tr.iter = 20000
stream = dataset.FrameStream(subj, device=dev, workers=3, prefetch=4)
cfg.perturb = cfg.train.perturb
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for block in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(250):
        loss, _ = tr.train_step(next(stream))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('train block %d: %.2f ms per iteration, loss %.5f, device %.2f GB (max %.2f), host max RSS %.0f MB'
          % (block, dt / 250 * 1e3, float(loss), torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, rss()), flush=True)
stream.close()
tr.grad_sync.finish()
net.eval()
cfg.perturb, cfg.amd.diagnostics = 0., False
out = tempfile.mkdtemp()
big = dataset.Subject(d)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = run.run_movement(net, big, render_folder_name='soak%d' % rep, logdir=out, device=dev, test_num=60)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('movement pass %d: 60 frames, %.2f ms per frame, psnr %.3f, guard hits %d, watched %d, device %.2f GB (max %.2f), host max RSS %.0f MB'
          % (rep, dt / 60 * 1e3, res['metrics']['psnr'], net.f16_range_hits, net.f16_range_watched,
             torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, rss()), flush=True)
