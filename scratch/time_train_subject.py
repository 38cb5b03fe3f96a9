"""Scratch: end-to-end training rate from a prepared subject directory (512x512 synthetic subject, 16 frames):
data side = dataset.FrameStream with the device-resident frame cache vs the numpy route.
    python scratch/time_train_subject.py [iters]"""
import os, pickle, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from humannerf_amd import dataset, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
from humannerf_amd.train import Trainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
d = tempfile.mkdtemp()
os.makedirs(d + '/images'); os.makedirs(d + '/masks')
H = W = 512
J = scene.TPOSE_JOINTS.astype(np.float64)
cams, infos = {}, {}
rs = np.random.RandomState(0)
for n in range(16):
    name = 'f%03d' % n
    K, E = scene.tpose_camera(np.array([W, H], dtype=np.float32), 4.0, 1250.0)
    cams[name] = {'intrinsics': K.astype(np.float64), 'extrinsics': E.astype(np.float64), 'distortions': np.zeros(5)}
    infos[name] = {'Rh': np.zeros(3), 'Th': np.zeros(3), 'poses': rs.randn(72) * 0.1, 'joints': J, 'tpose_joints': J}
    yy, xx = np.mgrid[0:H, 0:W]
    m = ((yy - H / 2) ** 2 / (H * 0.4) ** 2 + (xx - W / 2) ** 2 / (W * 0.2) ** 2 < 1)
    Image.fromarray((np.stack([m] * 3, -1) * 255).astype(np.uint8)).save(d + '/masks/' + name + '.png')
    Image.fromarray(rs.randint(0, 255, (H, W, 3)).astype(np.uint8)).save(d + '/images/' + name + '.png')
pickle.dump(cams, open(d + '/cameras.pkl', 'wb')); pickle.dump(infos, open(d + '/mesh_infos.pkl', 'wb'))
pickle.dump({'joints': J}, open(d + '/canonical_joints.pkl', 'wb'))

dev = torch.device('cuda:0')
cfg.train.lossweights.lpips, cfg.N_samples = 0.0, 128
subj = dataset.Subject(d)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()})
net = net.to(dev)
for cache in (True, False):
    tr = Trainer(net)
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
