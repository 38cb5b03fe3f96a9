"""Host cost of one training item of dataset.DeviceFrameCache (no training running): where the data side's time goes.
    python profiles/tools/time_items.py [lens]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, scene
from humannerf_amd.config import cfg
lens = len(sys.argv) > 1 and sys.argv[1] == 'lens'
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=8, size=1024 if lens else 512, binary_mask=True,
                              distortions=scene.ZJU_LIKE_DISTORTION if lens else None)
cfg.resize_img_scale = 0.5 if lens else 1.0
dev = torch.device('cuda:0')
subj = dataset.Subject(d)
cache = dataset.DeviceFrameCache(subj, dev)
for i in range(8):
    cache.train_batch(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    cache.train_batch(i % 8)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s: %.2f ms of host time per item (%.2f ms incl. the GPU work behind it)' % ('lens' if lens else 'plain', (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    cache.train_batch(i % 8)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue())
