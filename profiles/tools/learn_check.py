"""Does the whole training stack learn?  A synthetic subject (analytic images behind a lens, 1024^2 at scale 0.5), the
reference's initialisation, `iters` iterations of train_step from FrameStream, PSNR of run_movement against the frames
before and after.    python profiles/tools/learn_check.py [iters]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, run, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.train import Trainer
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(0); np.random.seed(0)
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=8, size=1024, distortions=scene.ZJU_LIKE_DISTORTION, radius=6.0)
cfg.resize_img_scale, cfg.bgcolor = 0.5, [0., 0., 0.]
cfg.train.lossweights.lpips, cfg.N_samples = 0.0, 128
dev = torch.device('cuda:0')
subj = dataset.Subject(d)
net = Network().to(dev)                                            # the reference's initialisation (network_util.py:163-290)
out = tempfile.mkdtemp()
def psnr(tag):
    net.eval(); cfg.perturb, cfg.amd.diagnostics = 0., False
    cfg.eval_iter = 10000000
    r = run.run_movement(net, subj, render_folder_name=tag, logdir=out, device=dev)
    net.train(); cfg.amd.diagnostics = True
    return r['metrics']['psnr']
p0 = psnr('before')
tr = Trainer(net)
stream = dataset.FrameStream(subj, device=dev, workers=3, prefetch=4, bgcolor=(0., 0., 0.))
cfg.perturb = cfg.train.perturb
t0 = time.perf_counter(); losses = []
for it in range(iters):
    loss, _ = tr.train_step(next(stream))
    if it % 250 == 0 or it == iters - 1:
        losses.append(float(loss)); print('iter %5d loss %.5f' % (tr.iter, losses[-1]), flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
stream.close(); tr.grad_sync.finish()
p1 = psnr('after')
print('PSNR of the 8 training frames: %.2f dB before, %.2f dB after %d iterations (%.1f s, %.2f ms per iteration)' % (p0, p1, iters, dt, dt / iters * 1e3))
