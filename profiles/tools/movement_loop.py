"""run.run_movement (frames loaded WITH their images: PNG decode, undistortion, resize, SMPL helpers, device ray
generation, render, unpack, metrics, PNG writer) against the pure render of the same frames' rays.
    python profiles/tools/movement_loop.py [n_frames] [lens]
``lens``: 1024x1024 PNGs with lens distortion, cfg.resize_img_scale = 0.5 (the ZJU-387 setting); rendered at 512x512."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, ops, run, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lens = len(sys.argv) > 2 and sys.argv[2] == 'lens'
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=n, size=1024 if lens else 512, binary_mask=True,
                              distortions=scene.ZJU_LIKE_DISTORTION if lens else None)
cfg.resize_img_scale = 0.5 if lens else 1.0
cfg.N_samples, cfg.perturb, cfg.amd.diagnostics = 128, 0., False
dev = torch.device('cuda:0')
subj = dataset.Subject(d)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()})
net = net.to(dev).eval()

# pure render: the rays of every frame resident on the device, nothing else in the loop
fr = [subj.movement_frame(i, image_size=(512, 512)) for i in range(n)]
keys = ('dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor')
items = []
for f in fr:
    g = ops.gen_rays(f['K'], f['E'], f['ray_bbox_min_xyz'], f['ray_bbox_max_xyz'], 512, 512, device=dev)
    dct = {k: torch.as_tensor(np.ascontiguousarray(f[k])).to(dev) for k in keys}
    dct.update(rays=g['rays'], near=g['near'], far=g['far'])
    items.append(dct)
with torch.no_grad():
    for it in items[:2]:
        net(**it, iter_val=1e7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in items:
        net(**it, iter_val=1e7)
    torch.cuda.synchronize(); pure = (time.perf_counter() - t0) / n
rays = float(np.mean([it['rays'].shape[1] for it in items]))
del items

out = tempfile.mkdtemp()
run.run_movement(net, subj, logdir=out, device=dev, test_num=2)                      # warm-up (writer threads, tables)
torch.cuda.synchronize(); t0 = time.perf_counter()
cfg.amd.loop_timing = True
res = run.run_movement(net, subj, render_folder_name='timed', logdir=out, device=dev)
torch.cuda.synchronize(); loop = (time.perf_counter() - t0) / n
from humannerf_amd import render
first = render.render_frames.last_prefetch['wait_ms'][0] * 1e-3            # pipeline fill: the first frame has to be built
steady = (loop * n - first) / n
print('%s: %d frames, %.0f rays per frame; pure render %.2f ms per frame; run_movement %.2f ms per frame (%.2f fps), '
      'loop / pure = %.3f; without the %.0f ms the first frame takes to build: %.2f ms per frame, loop / pure = %.3f'
      % ('lens (1024^2 PNGs, distortion, scale 0.5)' if lens else 'plain (512^2 PNGs)', n, rays,
         pure * 1e3, loop * 1e3, 1 / loop, loop / pure, first * 1e3, steady * 1e3, steady / pure), flush=True)
print('psnr', res['metrics'])
lp = render.render_frames.last_prefetch
print('prefetch: %d builders at the end; build ms median %.1f max %.1f; renderer waited ms median %.2f max %.1f' % (lp.get('workers', 0), np.median(lp['build_ms']), max(lp['build_ms']), np.median(lp['wait_ms']), max(lp['wait_ms'])))
print('main thread per frame (median ms): submit %.2f, wait for image %.2f, on_image %.2f' % tuple(np.median(lp[k]) for k in ('submit_ms', 'image_wait_ms', 'on_image_ms')))

if 'gpu_ms' in lp:
    print('GPU per frame (events on the render stream): busy median %.2f ms, gap to the next frame median %.2f max %.2f ms' % (np.median(lp['gpu_ms']), np.median(lp['gpu_gap_ms']), max(lp['gpu_gap_ms'])))
