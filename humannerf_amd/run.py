"""The render modes of the reference's run.py as functions over this package's pieces (the
callers on the far side of the path).  No command line: the reference's own run.py keeps working unchanged with
``network_module: 'humannerf_amd.network'``; these are for callers that want the whole loop on
the MI355X-native side -- dataset.Subject frames (camera only: rays are generated on the device), render.render_frames
(frame-sharded over the ranks, images unpacked and quantised on the device, asynchronous copies to pinned memory),
render.ImageWriter / MetricsWriter on worker threads.

  run_movement   run.py:212-445   every frame of the subject with its own camera and pose; render | truth (| alpha)
                                  side by side, PSNR (SSIM, LPIPS) per image and averaged
  run_freeview   run.py:67-170    one training frame seen from a camera orbiting the subject
  run_tpose      run.py:178-183   the canonical pose on a turntable, non-rigid motion off

Output layout as in the reference: ``<logdir>/<load_net><eval_output_tag>/<folder>/NAME.png`` plus
``<folder>-metrics.perimg.txt / .average.txt`` (movement) and the stacked frames (MP4 when imageio is importable).
With world > 1 every rank writes its own frames into the same folder; metrics are reduced by the caller
(``MetricsWriter`` files are per rank: ``<folder>.rank<r>``).
"""
import os

import numpy as np

from . import render
from .config import cfg


def _output_dir(logdir=None):
    return os.path.join(logdir if logdir is not None else cfg.get('logdir', '.'),
                        str(cfg.get('load_net', 'latest')) + str(cfg.get('eval_output_tag', '')))


class _Frames:
    """Sequence of per-frame input dicts built on demand (a movement sequence holds an image per frame)."""

    def __init__(self, n, make):
        self.n, self.make = n, make

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.make(i)


def _panels(rgb8, alpha8, truth8=None):
    """run.py:143-149 / 359-364: [render | truth if cfg.show_truth | alpha if cfg.show_alpha]."""
    imgs = [rgb8]
    if cfg.get('show_truth', False) and truth8 is not None:
        imgs.append(truth8)
    if cfg.get('show_alpha', False):
        imgs.append(alpha8)
    return np.concatenate(imgs, axis=1)


def _render_loop(network, frames, names, folder, logdir, rank, world, device, metrics=None):
    out_dir = _output_dir(logdir)
    writer = render.ImageWriter(out_dir, folder)
    own = list(range(rank, len(frames), world))

    def on_image(i, rgb8, alpha8, truth8=None):
        writer.append(_panels(rgb8, alpha8, truth8), img_name=names[i])
        if metrics is not None and truth8 is not None:
            metrics.append(name=names[i], pred=rgb8, target=truth8, mask=None)

    old = cfg.perturb
    cfg.perturb = 0.                                                   # run.py:71, 214
    try:
        images = render.render_frames(network, frames, rank=rank, world=world, device=device, on_image=on_image,
                                      show_truth=True)
    finally:
        cfg.perturb = old
    stack = writer.finalize()
    averages = metrics.finalize() if metrics is not None else None
    return {'frames': own, 'images': images, 'image_dir': writer.image_dir, 'stack': stack, 'metrics': averages}


def run_movement(network, subject, render_folder_name='movement', logdir=None, rank=0, world=1, device=None,
                 test_num=-1, metrics=None, lpips_fn=None):
    """run.py:212-445.  Frames are loaded with their images (truth panel and metrics), rays come from the device ray
    generator.  Returns the per-rank result dict of the loop; ``['metrics']`` holds this rank's averages."""
    cfg.show_truth = True
    device = device or next(network.parameters()).device
    n = len(subject) if test_num < 0 else min(test_num, len(subject))
    # camera-only frames: rays come from the device generator, the truth pixels are picked on the device
    # (the prefetcher thread of render_frames builds them: PNG decoding on the host, undistortion / composite / resize on
    # the device when there is one)
    frames = _Frames(n, lambda i: subject.movement_frame(i, load_image=True, device=device))
    names = [str(subject.framelist[i]).replace('/', '-') for i in range(n)]
    suffix = '' if world == 1 else '.rank%d' % rank
    mw = render.MetricsWriter(_output_dir(logdir), render_folder_name + suffix, dataset=subject.dataset_path,
                              metrics=metrics, lpips_fn=lpips_fn)
    return _render_loop(network, frames, names, render_folder_name, logdir, rank, world, device, metrics=mw)


def run_freeview(network, subject, frame_idx=None, total_frames=None, render_folder_name=None, logdir=None, rank=0,
                 world=1, device=None, image_size=None, src_type='zju_mocap'):
    """run.py:67-176 with data_type 'freeview' (freeview.py:172-280)."""
    frame_idx = int(cfg.get('freeview', {}).get('frame_idx', 0)) if frame_idx is None else int(frame_idx)
    total = int(cfg.get('render_frames', 100)) if total_frames is None else int(total_frames)
    if image_size is None:
        image_size = subject.image_size(subject.framelist_all[frame_idx])
    # background: cfg.bgcolor, like every non-train dataset the reference builds (create_dataset.py:40)
    frames = _Frames(total, lambda i: subject.freeview_frame(i, total, train_frame_idx=frame_idx, src_type=src_type,
                                                             image_size=image_size, bgcolor=cfg.bgcolor))
    folder = render_folder_name or cfg.get('render_folder_name', '') or 'freeview_%d' % frame_idx
    return _render_loop(network, frames, [None] * total if world == 1 else ['%06d' % i for i in range(total)], folder,
                        logdir, rank, world, device)


def run_tpose(network, subject, total_frames=None, render_folder_name=None, logdir=None, rank=0, world=1, device=None,
              image_size=None):
    """run.py:178-183: the turntable of tpose.py with cfg.ignore_non_rigid_motions = True."""
    total = int(cfg.get('render_frames', 100)) if total_frames is None else int(total_frames)
    frames = _Frames(total, lambda i: subject.tpose_frame(i, total, image_size=image_size, bgcolor=cfg.bgcolor))
    old = cfg.ignore_non_rigid_motions
    cfg.ignore_non_rigid_motions = True
    try:
        folder = render_folder_name or cfg.get('render_folder_name', '') or 'tpose'
        return _render_loop(network, frames, [None] * total if world == 1 else ['%06d' % i for i in range(total)],
                            folder, logdir, rank, world, device)
    finally:
        cfg.ignore_non_rigid_motions = old
