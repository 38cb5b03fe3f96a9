"""Fixture for the image side of the path: feeds seeded arrays through the REFERENCE's own
``run.unpack_to_image`` (run.py:48-65), ``to_8b_image`` / ``to_8b3ch_image`` / ``tile_images`` (core/utils/image_util.py:21-53),
``compute_psnr`` and ``MetricsWriter`` (core/utils/metrics_util.py:9-88) and stores inputs + outputs in
tests/golden/image_unpack.npz (+ the two metric text files as strings).

    python oracle/make_golden_images.py

Build container only.  Import recipe as in make_golden.py, plus empty stub modules for ``termcolor``, ``imageio``,
``skimage`` / ``skimage.metrics`` and ``tqdm`` if missing: they are imported at module top by the reference's
image_util / metrics_util / run.py but not called by the functions used here (PNG / MP4 encoding and SSIM are NOT
part of this fixture).
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
GOLD = os.path.join(REPO, 'tests', 'golden')


def main():
    import numpy as np
    import torch
    from make_golden import import_reference
    for name in ['termcolor', 'imageio', 'skimage', 'skimage.metrics']:
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    sys.modules['termcolor'].colored = getattr(sys.modules['termcolor'], 'colored', lambda s, *a, **k: s)
    if not hasattr(sys.modules['skimage.metrics'], 'structural_similarity'):
        sys.modules['skimage.metrics'].structural_similarity = None
        sys.modules['skimage'].metrics = sys.modules['skimage.metrics']
    cfg, _ = import_reference()
    import run as ref_run
    from core.utils import image_util as ref_img, metrics_util as ref_met

    rs = np.random.RandomState(314)
    H, W = 37, 53
    ray_mask = rs.rand(H * W) > 0.35
    n = int(ray_mask.sum())
    rgb = (rs.rand(n, 3) * 1.3 - 0.15).astype(np.float32)            # includes values outside [0, 1]: clipping
    alpha = (rs.rand(n) * 1.2 - 0.1).astype(np.float32)
    truth = rs.rand(n, 3).astype(np.float32)
    out = {'H': H, 'W': W, 'ray_mask': ray_mask, 'rgb': rgb, 'alpha': alpha, 'truth': truth}
    for tag, bg in (('black', np.array([0., 0., 0.])), ('white', np.array([255., 255., 255.])),
                    ('grey', np.array([30., 120., 250.]))):
        rgb_img, alpha_img, truth_img = ref_run.unpack_to_image(W, H, ray_mask, bg / 255., rgb.copy(), alpha.copy(),
                                                                truth.copy())
        out[tag + '_bg'] = bg.astype(np.float32)
        out[tag + '_rgb_img'], out[tag + '_alpha_img'], out[tag + '_truth_img'] = rgb_img, alpha_img, truth_img
        r2, _, t2 = ref_run.unpack_to_image(W, H, ray_mask, bg / 255., rgb.copy(), alpha.copy())
        assert np.array_equal(r2, rgb_img)
        out[tag + '_truth_none'] = np.asarray(t2)                     # without truth: the float32 background plane
    x = (rs.rand(9, 11) * 1.4 - 0.2).astype(np.float32)
    out['to8b_in'], out['to8b_out'], out['to8b3ch_out'] = x, ref_img.to_8b_image(x), ref_img.to_8b3ch_image(x)
    tiles = [rs.randint(0, 255, (6, 5, 3)).astype(np.uint8) for _ in range(10)]
    out['tiles_in'] = np.stack(tiles)
    out['tiles_out_4'] = ref_img.tile_images(tiles, imgs_per_row=4)
    out['tiles_out_3'] = ref_img.tile_images(tiles[:3], imgs_per_row=4)

    # PSNR (float images in 0..1, optional (H, W, 1) bool mask) and the MetricsWriter text files with metrics = psnr
    pred = torch.from_numpy(rs.rand(H, W, 3).astype(np.float32))
    target = torch.from_numpy(np.clip(pred.numpy() + rs.randn(H, W, 3).astype(np.float32) * 0.05, 0, 1))
    mask = torch.from_numpy(rs.rand(H, W, 1) > 0.5)
    out['psnr_pred'], out['psnr_target'], out['psnr_mask'] = pred.numpy(), target.numpy(), mask.numpy()
    out['psnr'] = np.float32(ref_met.compute_psnr(pred, target).item())
    out['psnr_masked'] = np.float32(ref_met.compute_psnr(pred, target, mask).item())
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        if 'eval' not in cfg:
            cfg.eval = type(cfg)()
        cfg.eval.metrics = ['psnr']
        mw = ref_met.MetricsWriter(td, 'movement', dataset='zju_387_test', lpips_computer=object())
        mw.append('frame_000000', (pred.numpy() * 255).astype(np.uint8).astype(np.float32), target.numpy() * 255.0)
        mw.append('frame_000001', pred.numpy(), target.numpy(), mask)
        mw.finalize()
        out['metrics_perimg_txt'] = np.array(open(os.path.join(td, 'movement-metrics.perimg.txt')).read())
        out['metrics_average_txt'] = np.array(open(os.path.join(td, 'movement-metrics.average.txt')).read())
    np.savez_compressed(os.path.join(GOLD, 'image_unpack.npz'), **out)
    print('wrote image_unpack.npz:', {k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})
    print(str(out['metrics_perimg_txt']), str(out['metrics_average_txt']))


if __name__ == '__main__':
    main()
