"""Device statement (csrc/hnrf_image.hip) of the OpenCV steps of Dataset.load_image against the host statement
(humannerf_amd/imageproc.py; itself pinned by tests/test_image_cpu.py -- cv2 is not importable, parity with OpenCV's
binaries unpinned).  Both follow one operation order in float64 without fused multiply-adds, so:
  * undistortion (uint8): bit for bit;  * composite + Lanczos-4 resize (float32 after / 255): bit for bit;
  * mask resize: bit for bit;           * DeviceFrameCache items = Subject.train_frame items on a distorted,
                                          half-scale subject (what every 387 / wild yaml configures)."""
import numpy as np
import pytest
import torch

from humannerf_amd import dataset, imageproc as ip, ops, scene
from humannerf_amd.config import cfg

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('shape,focal', [((1024, 1024, 3), 1100.0), ((301, 517, 3), 400.0), ((64, 80, 1), 90.0)])
def test_undistort_kernel_equals_host(shape, focal):
    H, W, C = shape
    rs = np.random.RandomState(H)
    img = rs.randint(0, 256, shape).astype(np.uint8)
    K = np.array([[focal, 0.0, W / 2 + 0.7], [0, focal * 0.99, H / 2 - 1.9], [0, 0, 1.]])
    for D in (np.array(scene.ZJU_LIKE_DISTORTION), np.array([0.31, -0.2, 2e-3, -1e-3, 0.05]), np.zeros(5)):
        want = ip.undistort_u8(img if C > 1 else img[:, :, 0], K, D)
        got = ops.undistort_image(_up(img), K, D).cpu().numpy()
        assert np.array_equal(got if C > 1 else got[:, :, 0], want), (shape, D.tolist())
        if not D.any():
            assert np.array_equal(got, img)                                   # D = 0: the identity, bit for bit
        else:
            assert (got != img).mean() > 0.5


@pytest.mark.parametrize('scale', [0.5, 0.37, 1.0, 1.25])
def test_composite_windows_kernel_equals_host(scale):
    rs = np.random.RandomState(7)
    Hs, Ws = 150, 212
    orig = rs.randint(0, 256, (Hs, Ws, 3)).astype(np.uint8)
    alpha = np.clip(rs.randint(-200, 456, (Hs, Ws, 3)), 0, 255).astype(np.uint8)      # plenty of exact 0 and 255
    bg = (rs.rand(3) * 255).astype(np.float32)
    img, a = ip.load_step(orig, alpha, bg, scale=scale)
    want = (img / 255.).astype(np.float32)
    Hd, Wd = want.shape[:2]
    full = ops.composite_windows(_up(orig), _up(alpha), _up(bg), [(0, 0)], Hd, Wd, scale=scale)[0].cpu().numpy()
    assert full.shape == want.shape and np.array_equal(full, want), np.abs(full - want).max()
    wins = [(0, 0), (Wd - 16, Hd - 16), (5, 9), (Wd - 16, 0)]
    got = ops.composite_windows(_up(orig), _up(alpha), _up(bg), wins, 16, 16, scale=scale).cpu().numpy()
    for g, (x0, y0) in zip(got, wins):
        assert np.array_equal(g, want[y0:y0 + 16, x0:x0 + 16])
    m = ops.resize_mask(_up(alpha), scale).cpu().numpy()
    assert np.array_equal(m, a[:, :, 0].astype(np.float32))
    with pytest.raises(Exception, match='leave'):
        ops.composite_windows(_up(orig), _up(alpha), _up(bg), [(Wd - 15, 0)], 16, 16, scale=scale)


def test_device_frame_cache_on_a_distorted_half_scale_subject(tmp_path):
    """with `distortions` on every camera and resize_img_scale 0.5 the cache route is taken and
    returns what the host route (Subject.train_frame) returns -- patches, masks, targets bit for bit, rays to the
    ray generator's 2e-6 -- and the whole-image device route equals load_image."""
    names = scene.write_synthetic_subject(str(tmp_path), n_frames=3, size=256, distortions=scene.ZJU_LIKE_DISTORTION)
    subj = dataset.Subject(str(tmp_path))
    old = (cfg.patch.N_patches, cfg.patch.size, cfg.get('resize_img_scale', 1.0))
    cfg.patch.N_patches, cfg.patch.size, cfg.resize_img_scale = 4, 16, 0.5
    try:
        cache = dataset.DeviceFrameCache(subj, DEV)
        for idx, seed in ((0, 1), (2, 2), (1, 3), (0, 4)):
            np.random.seed(seed)
            want = subj.train_frame(idx)
            np.random.seed(seed)
            got = cache.train_batch(idx)
            assert got['img_width'] == want['img_width'] == 128 and got['resize_parity'] == want['resize_parity'] == 'unpinned'
            for k in ('patch_masks', 'ray_mask', 'bgcolor', 'target_patches', 'target_rgbs'):
                assert np.array_equal(got[k].cpu().numpy(), want[k]), k
            assert np.array_equal(got['patch_div_indices'].numpy(), want['patch_div_indices'])
            for k in ('rays', 'near', 'far'):
                np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=2e-6, atol=2e-6, err_msg=k)
        bg = np.array([3., 200., 77.], dtype=np.float32)
        img, a, flag = subj.load_image(names[1], bg)
        dimg, dmask, dflag = subj.load_image_device(names[1], bg, DEV)
        assert flag == dflag == 'unpinned'
        assert np.array_equal(dimg.cpu().numpy(), (img / 255.).astype(np.float32))
        assert np.array_equal(dmask.cpu().numpy(), a[:, :, 0].astype(np.float32))
        # (last: the stream's worker threads draw from the global numpy generator)
        stream = dataset.FrameStream(subj, device=DEV, seed=3, workers=1)
        assert stream.cache is not None                                        # round 2 fell back to the numpy route here
        b = next(stream)
        stream.close()
        assert b['target_patches'].shape == (4, 16, 16, 3) and b['rays'].is_cuda
    finally:
        cfg.patch.N_patches, cfg.patch.size, cfg.resize_img_scale = old


def test_frame_stream_items_do_not_depend_on_the_threads(tmp_path):
    """FrameStream's default draws (cfg.amd.exact_patch_draws = False): every item has its own numpy Generator keyed by
    (seed, rank, ticket), so the sequence of batches is the same whatever the number of worker threads and however they
    interleave -- which the reference's global-generator draws, made from DataLoader workers, are not.  The exact draws
    stay available (and are what test_device_frame_cache_* compare with the host route)."""
    scene.write_synthetic_subject(str(tmp_path), n_frames=3, size=96)
    subj = dataset.Subject(str(tmp_path))
    old = (cfg.patch.N_patches, cfg.patch.size, cfg.get('resize_img_scale', 1.0))
    cfg.patch.N_patches, cfg.patch.size, cfg.resize_img_scale = 3, 16, 1.0
    try:
        runs = []
        for workers in (1, 3):
            st = dataset.FrameStream(subj, device=DEV, seed=11, workers=workers, prefetch=4)
            assert not st.exact_draws
            runs.append([next(st) for _ in range(7)])
            st.close()
        for a, b in zip(*runs):
            for k in ('rays', 'target_patches', 'patch_masks', 'bgcolor', 'near'):
                assert torch.equal(a[k], b[k]), k
        assert not torch.equal(runs[0][0]['bgcolor'], runs[0][1]['bgcolor'])         # random background per item (train.py:513-516)
        other = dataset.FrameStream(subj, device=DEV, seed=12, workers=1)
        assert not torch.equal(next(other)['bgcolor'], runs[0][0]['bgcolor'])
        other.close()
    finally:
        cfg.patch.N_patches, cfg.patch.size, cfg.resize_img_scale = old
