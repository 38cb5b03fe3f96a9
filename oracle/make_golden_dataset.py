"""Fixture for the on-disk formats and the per-frame input assembly: writes a small
synthetic subject directory in the layout of tools/prepare_zju_mocap/prepare_dataset.py:172-221 (cameras.pkl,
mesh_infos.pkl, canonical_joints.pkl by pickle.dump of dicts of numpy arrays, images/ + masks/ PNGs) under
tests/golden/subject_synth/, and the dict the REFERENCE's own numpy helpers (core/utils/body_util.py,
camera_util.py: body_pose_to_body_RTs, get_canonical_global_tfms, approx_gaussian_bone_volumes, get_rays_from_KRT,
rays_intersect_3d_bbox -- they import under the cv2 stub) produce for every frame, in
tests/golden/subject_synth_expected.npz.

    python oracle/make_golden_dataset.py

Not pinned (cv2 absent, see humannerf_amd/dataset.py): the Rodrigues step of apply_global_tfm_to_camera -- the
extrinsics handed to the reference helpers here come from dataset.apply_global_tfm_to_camera -- and the two image
steps cv2.undistort / cv2.resize (humannerf_amd/imageproc.py; tests/test_image_cpu.py pins them by properties).  For a
1024x1024 subject with lens distortion at resize_img_scale 0.5 the camera side IS pinned here ('lens/*').  No SMPL model is
involved: joints are the synthetic skeleton of scene.py plus seeded noise."""
import os
import pickle
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
OUT = os.path.join(REPO, 'tests', 'golden', 'subject_synth')
H, W = 64, 48


def main():
    import numpy as np
    from PIL import Image
    from make_golden import import_reference
    cfg, _ = import_reference()
    from core.utils import body_util as rb, camera_util as rc
    from humannerf_amd import dataset, scene

    rs = np.random.RandomState(77)
    os.makedirs(os.path.join(OUT, 'images'), exist_ok=True)
    os.makedirs(os.path.join(OUT, 'masks'), exist_ok=True)
    J = scene.TPOSE_JOINTS.astype(np.float64) + rs.randn(24, 3) * 0.004
    cameras, mesh_infos, expected = {}, {}, {}
    names = ['frame_%06d' % i for i in (3, 10, 42)]
    for n, name in enumerate(names):
        poses = rs.randn(72) * 0.25
        Rh = poses[:3].copy() * (0.0 if n == 0 else 1.0)
        poses[:3] = 0 if n < 2 else poses[:3]
        Th = rs.randn(3) * 0.1
        joints = J + rs.randn(24, 3) * 0.03                     # stands for the posed skeleton (SMPL is licence-gated)
        K, E = scene.tpose_camera(np.array([W, H], dtype=np.float32), 4.0 + n, 150.0 + 10 * n)
        E = E.astype(np.float64)
        E[:3, 3] += rs.randn(3) * 0.05
        cameras[name] = {'intrinsics': K.astype(np.float64), 'extrinsics': E, 'distortions': np.zeros(5)}
        mesh_infos[name] = {'Rh': Rh, 'Th': Th, 'poses': poses, 'joints': joints, 'tpose_joints': J + rs.randn(24, 3) * 0.002}
        # a blob as the subject mask and a seeded image
        yy, xx = np.mgrid[0:H, 0:W]
        m = ((yy - H / 2) ** 2 / (H * 0.35) ** 2 + (xx - W / 2) ** 2 / (W * 0.22) ** 2 < 1.0)
        Image.fromarray((np.stack([m] * 3, -1) * 255).astype(np.uint8)).save(os.path.join(OUT, 'masks', name + '.png'))
        Image.fromarray(rs.randint(0, 255, (H, W, 3)).astype(np.uint8)).save(os.path.join(OUT, 'images', name + '.png'))
    with open(os.path.join(OUT, 'cameras.pkl'), 'wb') as f:
        pickle.dump(cameras, f)
    with open(os.path.join(OUT, 'mesh_infos.pkl'), 'wb') as f:
        pickle.dump(mesh_infos, f)
    with open(os.path.join(OUT, 'canonical_joints.pkl'), 'wb') as f:
        pickle.dump({'joints': J}, f)

    # expected per-frame dict through the reference's helpers (train.py:481-755 in 'image' mode)
    cj = J.astype('float32')
    off = cfg.bbox_offset
    cmn, cmx = cj.min(0) - off, cj.max(0) + off
    expected['motion_weights_priors'] = rb.approx_gaussian_bone_volumes(cj, cmn, cmx, grid_size=cfg.mweight_volume.volume_size).astype('float32')
    expected['cnl_gtfms'] = rb.get_canonical_global_tfms(cj)
    expected['cnl_bbox_min_xyz'], expected['cnl_bbox_max_xyz'] = cmn.astype('float32'), cmx.astype('float32')
    expected['cnl_bbox_scale_xyz'] = 2.0 / (cmx.astype('float32') - cmn.astype('float32'))
    for name in names:
        info, cam = mesh_infos[name], cameras[name]
        poses = info['poses'].astype('float32')
        Rs, Ts = rb.body_pose_to_body_RTs(poses, info['tpose_joints'].astype('float32'))
        bbox = {'min_xyz': info['joints'].min(0) - off, 'max_xyz': info['joints'].max(0) + off}
        Kf = cam['intrinsics'][:3, :3].copy()
        Kf[:2] *= 1.0
        Ef = dataset.apply_global_tfm_to_camera(cam['extrinsics'], info['Rh'].astype('float32'), info['Th'].astype('float32'))
        ro, rd = rc.get_rays_from_KRT(H, W, Kf, Ef[:3, :3], Ef[:3, 3])
        ro, rd = ro.reshape(-1, 3).copy(), rd.reshape(-1, 3).copy()
        near, far, mask = rc.rays_intersect_3d_bbox(bbox, ro, rd)
        expected[name + '/dst_Rs'], expected[name + '/dst_Ts'] = Rs, Ts
        expected[name + '/dst_posevec'] = poses[3:] + 1e-2
        expected[name + '/E'] = Ef
        expected[name + '/rays'] = np.stack([ro[mask], rd[mask], rd[mask]], 0).astype('float32')
        expected[name + '/near'], expected[name + '/far'] = near[:, None].astype('float32'), far[:, None].astype('float32')
        expected[name + '/ray_mask'] = mask
        print(name, 'rays', int(mask.sum()), 'of', H * W)
    # run.py's 'tpose' mode (core/data/human_nerf/tpose.py:127-228) through the reference's helpers, 64x64 image;
    # the two cv2.Rodrigues conversions of the root rotation come from dataset.py (cv2 absent)
    total = 8
    for idx in (0, 3, 5):
        angle = 2 * np.pi / total * idx
        add = dataset.rodrigues_cv(np.array([0, -angle, 0], dtype='float32'))
        poses = np.zeros(72, dtype='float32')
        poses[:3] = dataset.rotation_to_rvec(add)
        x, y, z = 0., -0.25, 6.0
        campos = np.array([x, y, z], dtype='float32')
        camrot = rc.get_camrot(campos, lookat=np.array([0, y, 0.]), inv_camera=True)
        Et = np.eye(4, dtype='float32')
        Et[:3, :3] = camrot
        Et[:3, 3] = -camrot.dot(campos)
        Kt = np.eye(3, dtype='float32')
        Kt[0, 0] = Kt[1, 1] = 100.                                # (1250 in tpose.py: at 64 px every ray would hit)
        Kt[:2, 2] = 64 / 2.
        mnx, mxx = cmn, cmx
        pts = np.array([[mnx[0], mnx[1], mnx[2]], [mnx[0], mnx[1], mxx[2]], [mnx[0], mxx[1], mnx[2]], [mnx[0], mxx[1], mxx[2]],
                        [mxx[0], mnx[1], mnx[2]], [mxx[0], mnx[1], mxx[2]], [mxx[0], mxx[1], mnx[2]], [mxx[0], mxx[1], mxx[2]]]).dot(add)
        bbox = {'min_xyz': pts.min(0), 'max_xyz': pts.max(0)}
        ro, rd = rc.get_rays_from_KRT(64, 64, Kt, Et[:3, :3], Et[:3, 3])
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        near, far, mask = rc.rays_intersect_3d_bbox(bbox, ro, rd)
        Rs, Ts = rb.body_pose_to_body_RTs(poses, cj)
        key = 'tpose%d/' % idx
        expected[key + 'dst_Rs'], expected[key + 'dst_Ts'] = Rs, Ts
        expected[key + 'rays'] = np.stack([ro[mask], rd[mask], rd[mask]], 0).astype('float32')
        expected[key + 'near'], expected[key + 'far'] = near[:, None].astype('float32'), far[:, None].astype('float32')
        expected[key + 'ray_mask'] = mask
        print('tpose', idx, 'rays', int(mask.sum()))
    # ---- the ZJU-387 setting: 1024x1024 images with lens distortion, cfg.resize_img_scale = 0.5 (387/adventure.yaml:37).
    # The directory itself is NOT committed (scene.write_synthetic_subject regenerates it from seeds in the tests: the
    # two OpenCV steps cannot be run by the reference here, cv2 is absent); what the reference's helpers CAN pin is the
    # camera side of such a frame -- intrinsics scaled by the resize factor (train.py:560), rays of the 512x512 image,
    # bbox test -- stored as the packed hit mask and every 1024th kept ray.
    import tempfile
    lens_dir = tempfile.mkdtemp()
    lens_names = scene.write_synthetic_subject(lens_dir, n_frames=1, size=1024, distortions=scene.ZJU_LIKE_DISTORTION, seed=5, radius=6.0)
    with open(os.path.join(lens_dir, 'cameras.pkl'), 'rb') as f:
        lcam = pickle.load(f)[lens_names[0]]
    with open(os.path.join(lens_dir, 'mesh_infos.pkl'), 'rb') as f:
        linfo = pickle.load(f)[lens_names[0]]
    Kl = lcam['intrinsics'][:3, :3].copy()
    Kl[:2] *= 0.5
    El = dataset.apply_global_tfm_to_camera(lcam['extrinsics'], linfo['Rh'].astype('float32'), linfo['Th'].astype('float32'))
    ro, rd = rc.get_rays_from_KRT(512, 512, Kl, El[:3, :3], El[:3, 3])
    ro, rd = ro.reshape(-1, 3).copy(), rd.reshape(-1, 3).copy()
    lbbox = {'min_xyz': linfo['joints'].min(0) - off, 'max_xyz': linfo['joints'].max(0) + off}
    near, far, mask = rc.rays_intersect_3d_bbox(lbbox, ro, rd)
    pick = np.arange(0, int(mask.sum()), 1024)
    expected['lens/ray_mask_bits'] = np.packbits(mask)
    expected['lens/pick'] = pick
    expected['lens/rays_o'], expected['lens/rays_d'] = ro[mask][pick].astype('float32'), rd[mask][pick].astype('float32')
    expected['lens/near'], expected['lens/far'] = near[pick].astype('float32'), far[pick].astype('float32')
    print('lens frame: rays', int(mask.sum()), 'of', 512 * 512)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'subject_synth_expected.npz'), **expected)
    # the loader must refuse anything that is not plain data
    evil = os.path.join(OUT, 'not_data.pkl')
    with open(evil, 'wb') as f:
        pickle.dump({'x': os.path.join}, f)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
