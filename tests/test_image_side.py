"""The image side of the path against the REFERENCE's own functions: the fixture
tests/golden/image_unpack.npz holds what run.unpack_to_image, to_8b_image, to_8b3ch_image, tile_images, compute_psnr and
MetricsWriter of the reference returned for seeded inputs (oracle/make_golden_images.py).  render.unpack_to_image is
a torch restatement (device-side in production): byte-for-byte equality is asked, on the CPU here and on the GPU in
tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from humannerf_amd import render


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'image_unpack.npz'))


def check_unpack(g, device):
    H, W = int(g['H']), int(g['W'])
    T = lambda a: torch.from_numpy(np.asarray(a)).to(device)
    for tag in ('black', 'white', 'grey'):
        bg = g[tag + '_bg'] / 255.
        rgb8, a8, t8 = render.unpack_to_image(W, H, T(g['ray_mask']), bg, T(g['rgb']), T(g['alpha']), T(g['truth']))
        assert rgb8.dtype == torch.uint8 and tuple(rgb8.shape) == (H, W, 3)
        assert np.array_equal(rgb8.cpu().numpy(), g[tag + '_rgb_img'])
        assert np.array_equal(a8.cpu().numpy(), g[tag + '_alpha_img'])
        assert np.array_equal(t8.cpu().numpy(), g[tag + '_truth_img'])
        _, _, plane = render.unpack_to_image(W, H, T(g['ray_mask']), bg, T(g['rgb']), T(g['alpha']))
        assert np.array_equal(plane.cpu().numpy(), g[tag + '_truth_none'])


def test_unpack_to_image_matches_reference_bytes(g):
    check_unpack(g, torch.device('cpu'))


def test_8bit_and_tiling_match_reference(g):
    assert np.array_equal(render.to_8b_image(g['to8b_in']), g['to8b_out'])
    assert np.array_equal(render.to_8b3ch_image(g['to8b_in']), g['to8b3ch_out'])
    assert np.array_equal(render.to_8b_image(torch.from_numpy(g['to8b_in'])).numpy(), g['to8b_out'])
    tiles = list(g['tiles_in'])
    assert np.array_equal(render.tile_images(tiles, 4), g['tiles_out_4'])
    assert np.array_equal(render.tile_images(tiles[:3], 4), g['tiles_out_3'])


def test_psnr_and_metrics_writer_match_reference(g, tmp_path):
    pred, target, mask = torch.from_numpy(g['psnr_pred']), torch.from_numpy(g['psnr_target']), torch.from_numpy(g['psnr_mask'])
    assert abs(float(render.psnr(pred, target)) - float(g['psnr'])) <= 1e-5
    assert abs(float(render.psnr(pred, target, mask)) - float(g['psnr_masked'])) <= 1e-5
    mw = render.MetricsWriter(str(tmp_path), 'movement', dataset='zju_387_test', metrics=['psnr'])
    mw.append('frame_000000', (g['psnr_pred'] * 255).astype(np.uint8).astype(np.float32), g['psnr_target'] * 255.0)
    mw.append('frame_000001', g['psnr_pred'], g['psnr_target'], mask)
    avg = mw.finalize()
    assert open(tmp_path / 'movement-metrics.perimg.txt').read() == str(g['metrics_perimg_txt'])
    assert open(tmp_path / 'movement-metrics.average.txt').read() == str(g['metrics_average_txt'])
    assert abs(avg['psnr'] - 26.3092) < 1e-3
    with pytest.raises(ValueError):
        render.MetricsWriter(str(tmp_path), 'x', dataset='d', metrics=['lpips'])


def test_ssim_properties():
    """SSIM is unpinned (skimage absent): identities and monotonicity only."""
    rs = np.random.RandomState(0)
    a = rs.rand(40, 32, 3)
    assert abs(render.ssim(a, a) - 1.0) < 1e-12
    n1, n2 = np.clip(a + rs.randn(*a.shape) * 0.02, 0, 1), np.clip(a + rs.randn(*a.shape) * 0.2, 0, 1)
    assert 1.0 > render.ssim(a, n1) > render.ssim(a, n2) > 0.0
    m = np.zeros((40, 32), bool)
    m[5:30, 4:20] = True
    assert abs(render.ssim(a, a, m) - 1.0) < 1e-12


def test_image_writer_threads(tmp_path):
    from PIL import Image
    w = render.ImageWriter(str(tmp_path), 'freeview', workers=3)
    rs = np.random.RandomState(1)
    imgs = [rs.randint(0, 255, (24, 31, 3)).astype(np.uint8) for _ in range(9)]
    for i, im in enumerate(imgs):
        idx, name = w.append(im, img_name=None if i % 2 else 'named_%02d' % i)
        assert idx == i
    stack = w.finalize()
    files = sorted(os.listdir(tmp_path / 'freeview'))
    assert len(files) == 9 and '000001.png' in files and 'named_00.png' in files
    assert np.array_equal(np.asarray(Image.open(tmp_path / 'freeview' / 'named_04.png')), imgs[4])
    assert np.load(stack).shape == (9, 24, 31, 3)


def test_point_cloud_dumps(tmp_path):
    """append_3d / append_cnl_3d: the .obj text of image_util.py:85-109 (line format checked literally)."""
    w = render.ImageWriter(str(tmp_path), 'mv', workers=1, keep_frames=False)
    w.append(np.zeros((4, 5, 3), np.uint8))
    pts = np.arange(4 * 5 * 3, dtype=np.float32).reshape(4, 5, 3) / 7
    mask = np.zeros((4, 5), bool)
    mask[1, 2] = mask[3, 0] = True
    w.append_3d(pts, mask, depth_img=np.ones((4, 5), np.float32))
    w.append_cnl_3d(np.array([[0.5, -1.0, 2.0]]), np.array([[0.1, 0.2, 0.3]]))
    w.finalize()
    lines = open(tmp_path / 'mv_3d' / '000000.obj').read().splitlines()
    assert lines == ['v %.7f %.7f %.7f' % tuple(pts[1, 2]), 'v %.7f %.7f %.7f' % tuple(pts[3, 0])]
    assert open(tmp_path / 'mv_3d' / '000000-cnl.obj').read() == 'v 0.5000000 -1.0000000 2.0000000 0.1000000 0.2000000 0.3000000 \n'
    assert np.load(tmp_path / 'mv_3d' / '000000-depth.npy').shape == (4, 5)
