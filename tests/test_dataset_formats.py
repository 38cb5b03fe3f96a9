"""On-disk formats + per-frame input assembly against the fixture of
oracle/make_golden_dataset.py: a subject directory in the reference's layout (cameras.pkl / mesh_infos.pkl /
canonical_joints.pkl written by pickle.dump of numpy dicts like tools/prepare_zju_mocap/prepare_dataset.py:201-221,
images/, masks/) and the per-frame dict the REFERENCE's own body_util / camera_util helpers produce for it."""
import os
import pickle

import numpy as np
import pytest
import torch

from humannerf_amd import dataset

NAMES = ['frame_000003', 'frame_000010', 'frame_000042']
H, W = 64, 48


@pytest.fixture(scope='module')
def subj(golden_dir):
    return dataset.Subject(os.path.join(golden_dir, 'subject_synth'))


@pytest.fixture(scope='module')
def want(golden_dir):
    return np.load(os.path.join(golden_dir, 'subject_synth_expected.npz'))


def test_subject_directory_loads(subj, want):
    assert subj.framelist == NAMES and len(subj) == 3
    assert set(subj.cameras[NAMES[0]]) == {'intrinsics', 'extrinsics', 'distortions'}
    assert set(subj.mesh_infos[NAMES[0]]) == {'Rh', 'Th', 'poses', 'joints', 'tpose_joints', 'bbox'}
    assert np.allclose(subj.motion_weights_priors, want['motion_weights_priors'], atol=2e-6)
    assert np.allclose(subj.cnl_gtfms, want['cnl_gtfms'], atol=1e-7)


@pytest.mark.parametrize('name', NAMES)
def test_movement_frame_matches_reference_helpers(subj, want, name):
    """The numpy route (host_rays=True): every entry of the Network.forward dict equals what the reference's helpers
    give for the same record -- rays / near / far / ray_mask bit for bit."""
    fr = subj.movement_frame(NAMES.index(name), bgcolor=(0., 0., 0.), host_rays=True, image_size=(H, W))
    for k in ('rays', 'near', 'far', 'ray_mask'):
        assert np.array_equal(fr[k], want[name + '/' + k]), k
    assert np.allclose(fr['dst_Rs'], want[name + '/dst_Rs'], atol=1e-6) and np.array_equal(fr['dst_Ts'], want[name + '/dst_Ts'])
    assert np.array_equal(fr['dst_posevec'], want[name + '/dst_posevec'])
    for k in ('cnl_bbox_min_xyz', 'cnl_bbox_max_xyz', 'cnl_bbox_scale_xyz'):
        assert np.array_equal(fr[k], want[k]), k
    assert fr['rays'].shape[0] == 3 and fr['rays'].dtype == np.float32 and fr['near'].shape[1] == 1
    # the default (camera-only) form carries what the device-side ray generator needs instead
    cam = subj.movement_frame(NAMES.index(name), image_size=(H, W))
    assert 'rays' not in cam and np.allclose(cam['E'], want[name + '/E'], atol=1e-6)
    assert np.array_equal(cam['ray_bbox_min_xyz'], (subj.mesh_infos[name]['joints'].min(0) - 0.3).astype('float32'))
    assert cam['motion_weights_priors'] is subj.motion_weights_priors           # resident by identity across frames


def test_train_frame_patches(subj):
    """train.py:481-631: image + mask from the PNGs, six windows, rays of the windows only, targets cut from the
    composited image, patch masks; a seeded global generator gives the same patches."""
    from humannerf_amd.config import cfg
    old = (cfg.patch.size, cfg.patch.N_patches)
    cfg.patch.size, cfg.patch.N_patches = 16, 4
    try:
        np.random.seed(5)
        a = subj.train_frame(1, bgcolor=(10., 20., 30.))
        np.random.seed(5)
        b = subj.train_frame(1, bgcolor=(10., 20., 30.))
    finally:
        cfg.patch.size, cfg.patch.N_patches = old
    assert a['target_patches'].shape == (4, 16, 16, 3) and a['patch_masks'].shape == (4, 16, 16)
    n = int(a['patch_div_indices'][-1])
    assert a['rays'].shape == (3, n, 3) and a['near'].shape == (n, 1) and a['target_rgbs'].shape == (n, 3)
    assert int(a['patch_masks'].sum()) == n and a['resize_parity'] == 'exact'
    for k in ('rays', 'target_patches', 'patch_masks', 'patch_div_indices'):
        assert np.array_equal(a[k], b[k]), k
    # the rays' colours are the target patches' pixels under the masks, in order
    got = np.concatenate([a['target_patches'][i][a['patch_masks'][i]] for i in range(4)], 0)
    assert np.array_equal(got, a['target_rgbs'])
    # composite over the background: outside the mask blob the image IS the background
    assert np.allclose(a['target_patches'].min(), min(10. / 255., a['target_patches'].min()))


def test_freeview_frame_orbits(subj):
    f0 = subj.freeview_frame(0, 8, image_size=(H, W))
    f4 = subj.freeview_frame(4, 8, image_size=(H, W))
    f8 = subj.freeview_frame(8, 8, image_size=(H, W))
    assert np.allclose(f0['E'], f8['E'], atol=1e-5) and not np.allclose(f0['E'], f4['E'], atol=1e-2)
    for f in (f0, f4):
        R = f['E'][:3, :3].astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    assert tuple(f0['bgcolor']) == (255., 255., 255.)                       # freeview.py:74


@pytest.mark.parametrize('idx', [0, 3, 5])
def test_tpose_frame_matches_reference_helpers(subj, want, idx):
    """run.py's 'tpose' mode (tpose.py:127-228): zero pose turned about the vertical axis, fixed camera; rays / near /
    far / mask and the bone transforms against the reference's numpy helpers (fixture: oracle/make_golden_dataset.py)."""
    subj.TPOSE_CAM_PARAMS = {'radius': 6.0, 'focal': 100.}
    fr = subj.tpose_frame(idx, 8, host_rays=True, image_size=64)
    key = 'tpose%d/' % idx
    assert np.array_equal(fr['ray_mask'], want[key + 'ray_mask']) and 0 < fr['ray_mask'].sum() < 64 * 64
    for k in ('rays', 'near', 'far', 'dst_Rs', 'dst_Ts'):
        assert fr[k].shape == want[key + k].shape
        assert np.abs(fr[k] - want[key + k]).max() <= 2e-6, k
    assert np.allclose(fr['dst_posevec'], 0.01) and fr['bgcolor'].tolist() == [255., 255., 255.]
    assert np.array_equal(fr['cnl_bbox_min_xyz'], want['cnl_bbox_min_xyz'])
    # camera-only form for the device ray generator carries the rotated box
    cam = subj.tpose_frame(idx, 8, image_size=64)
    assert 'rays' not in cam and cam['K'][0, 0] == 100. and cam['ray_bbox_min_xyz'].shape == (3,)


def test_rotation_to_rvec_inverts_rodrigues():
    from humannerf_amd import dataset
    rs = np.random.RandomState(5)
    for v in [rs.randn(3) * s for s in (1e-6, 0.1, 1.0, 2.0)] + [np.array([0, np.pi, 0.]), np.array([0, -2.5, 0.]), np.zeros(3)]:
        R = dataset.rodrigues_cv(v)
        back = dataset.rotation_to_rvec(R)
        assert np.linalg.norm(back) <= np.pi + 1e-9
        assert np.abs(dataset.rodrigues_cv(back) - R).max() <= 1e-9


def test_rodrigues_and_global_transform():
    """cv2 is absent: closed-form properties only (parity with OpenCV's numerics unpinned)."""
    rs = np.random.RandomState(3)
    for _ in range(5):
        r = rs.randn(3)
        R = dataset.rodrigues_cv(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert np.allclose(R @ r, r, atol=1e-12)                            # the axis is fixed
        assert abs(np.trace(R) - (1 + 2 * np.cos(np.linalg.norm(r)))) < 1e-12
    assert np.array_equal(dataset.rodrigues_cv(np.zeros(3)), np.eye(3))
    E = np.eye(4)
    E[:3, 3] = [0.1, -0.2, 3.0]
    Th = np.array([0.3, 0.1, -0.2])
    out = dataset.apply_global_tfm_to_camera(E, np.zeros(3), Th)
    assert np.allclose(out[:3, 3], E[:3, 3] + Th) and np.allclose(out[:3, :3], np.eye(3))


def test_pickle_reader_refuses_code(golden_dir, tmp_path):
    with pytest.raises(pickle.UnpicklingError):
        dataset.load_array_pickle(os.path.join(golden_dir, 'subject_synth', 'not_data.pkl'))

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    p = tmp_path / 'evil.pkl'
    p.write_bytes(pickle.dumps({'a': Evil()}))
    with pytest.raises(pickle.UnpicklingError):
        dataset.load_array_pickle(str(p))
    ok = {'k': np.arange(6, dtype=np.float32).reshape(2, 3), 'n': np.float64(2.5), 'l': [1, 'x', (2, 3)]}
    back = dataset.load_array_pickle(pickle.dumps(ok))
    assert np.array_equal(back['k'], ok['k']) and back['n'] == 2.5 and back['l'] == ok['l']


def test_distorted_cameras_load(subj):
    """Round 2 refused such frames; cv2.undistort is restated now (imageproc.py, tests/test_image_cpu.py).  Lens models
    beyond the 5-vector the ZJU preparer writes (prepare_dataset.py:172-176) are refused by name."""
    name = NAMES[0]
    old = subj.cameras[name]['distortions']
    try:
        subj.cameras[name]['distortions'] = np.array([0.1, 0, 0, 0, 0.])
        img, alpha, flag = subj.load_image(name, np.zeros(3, dtype=np.float32))
        assert flag == 'unpinned' and img.shape == (H, W, 3)
        subj.cameras[name]['distortions'] = np.array([0.1, 0, 0, 0, 0., 0.2, 0, 0])
        with pytest.raises(NotImplementedError, match='rational'):
            subj.load_image(name, np.zeros(3, dtype=np.float32))
    finally:
        subj.cameras[name]['distortions'] = old


def test_checkpoint_tar_reader(tmp_path):
    """trainer.py:356-377 layout read back with weights_only=True; foreign pickles are refused by torch itself."""
    from humannerf_amd.train import load_checkpoint
    sd = {'cnl_mlp.module.pts_linears.0.weight': torch.randn(4, 3)}
    path = str(tmp_path / 'latest.tar')
    torch.save({'iter': 7, 'network': sd, 'optimizer': {'state': {}, 'param_groups': []}}, path)
    ck = load_checkpoint(path)
    assert ck['iter'] == 7 and torch.equal(ck['network']['cnl_mlp.module.pts_linears.0.weight'], sd['cnl_mlp.module.pts_linears.0.weight'])
    torch.save({'something': 1}, path)
    with pytest.raises(ValueError):
        load_checkpoint(path)


def test_frame_stream_shards_and_shuffles(subj):
    """FrameStream: every epoch is one permutation shared by the ranks, rank r takes every world-th element (wrapping,
    DistributedSampler's rule); batches arrive in that order whatever the worker threads do."""
    from humannerf_amd import dataset
    from humannerf_amd.config import cfg
    old = (cfg.patch.N_patches, cfg.patch.size)
    cfg.patch.N_patches, cfg.patch.size = 2, 16
    try:
        streams = [dataset.FrameStream(subj, rank=r, world=2, seed=5, workers=3, bgcolor=(0., 0., 0.)) for r in range(2)]
        got = [[next(s)['frame_name'] for _ in range(4)] for s in streams]
        for s in streams:
            s.close()
    finally:
        cfg.patch.N_patches, cfg.patch.size = old
    names = subj.framelist
    for epoch in range(2):
        perm = [names[i] for i in np.random.RandomState(5 + epoch).permutation(3)]
        padded = perm + perm[:1]                                          # 3 frames over 2 ranks: wrap to 4
        assert got[0][2 * epoch:2 * epoch + 2] == padded[0::2]
        assert got[1][2 * epoch:2 * epoch + 2] == padded[1::2]
    one = dataset.FrameStream(subj, seed=1, bgcolor=(0., 0., 0.))
    b = next(one)
    one.close()
    assert b['rays'].shape[0] == 3 and b['target_patches'].ndim == 4 and b['patch_masks'].dtype == bool


def test_freeview_and_tpose_renders_use_the_configured_background(subj, monkeypatch):
    """the reference builds every non-train dataset with bgcolor = cfg.bgcolor (create_dataset.py:40), so its
    free-viewpoint and T-pose renders sit on the configured background ([0, 0, 0] in the ZJU yamls), not on the datasets'
    white default.  run_freeview / run_tpose hand cfg.bgcolor to the frames they build."""
    from humannerf_amd import run
    from humannerf_amd.config import cfg
    seen = {}

    def fake_loop(network, frames, names, folder, logdir, rank, world, device, metrics=None):
        seen[folder] = [frames[i]['bgcolor'].tolist() for i in (0, len(frames) - 1)]
        return {}
    monkeypatch.setattr(run, '_render_loop', fake_loop)
    old = cfg.bgcolor
    try:
        cfg.bgcolor = [0., 0., 0.]
        run.run_freeview(None, subj, frame_idx=0, total_frames=4, image_size=(H, W))
        run.run_tpose(None, subj, total_frames=4, image_size=64)
        assert seen['freeview_0'] == [[0., 0., 0.]] * 2 and seen['tpose'] == [[0., 0., 0.]] * 2
        cfg.bgcolor = [255., 128., 0.]
        run.run_freeview(None, subj, frame_idx=1, total_frames=4, image_size=(H, W))
        assert seen['freeview_1'][0] == [255., 128., 0.]
    finally:
        cfg.bgcolor = old


def test_half_scale_camera_matches_reference_helpers(want, tmp_path):
    """The ZJU-387 setting (1024x1024 images, lens distortion, cfg.resize_img_scale = 0.5): the camera side of a frame --
    intrinsics scaled by the resize factor (train.py:560), image size by cvRound, rays of the 512x512 image, bbox test --
    against the reference's own numpy helpers (oracle/make_golden_dataset.py, 'lens/*': packed hit mask + every 1024th
    kept ray).  The subject directory is regenerated from its seeds (scene.write_synthetic_subject); its image side is
    what tests/test_image_cpu.py pins."""
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    names = scene.write_synthetic_subject(str(tmp_path), n_frames=1, size=1024, distortions=scene.ZJU_LIKE_DISTORTION,
                                          seed=5, radius=6.0)
    s = dataset.Subject(str(tmp_path))
    old = cfg.get('resize_img_scale', 1.0)
    try:
        cfg.resize_img_scale = 0.5
        assert s.image_size(names[0]) == (512, 512)
        fr = s.movement_frame(0, bgcolor=(0., 0., 0.), host_rays=True, image_size=s.image_size(names[0]))
    finally:
        cfg.resize_img_scale = old
    mask = np.unpackbits(want['lens/ray_mask_bits'])[:512 * 512].astype(bool)
    assert np.array_equal(fr['ray_mask'], mask) and 0.5 < mask.mean() < 0.99
    pick = want['lens/pick']
    assert np.array_equal(fr['rays'][0][pick], want['lens/rays_o']) and np.array_equal(fr['rays'][1][pick], want['lens/rays_d'])
    assert np.array_equal(fr['near'][pick, 0], want['lens/near']) and np.array_equal(fr['far'][pick, 0], want['lens/far'])
