"""On-disk formats of the reference and the per-frame input assembly in front of the path
(SURVEY.md section 8(f) rank 4 + the caller side of rank 1).

A prepared subject directory (tools/prepare_zju_mocap/prepare_dataset.py:172-221, tools/prepare_wild/
prepare_dataset.py:82-111) holds

    cameras.pkl           {frame: {'intrinsics' (3,3), 'extrinsics' (4,4) [, 'distortions' (5,)]}}
    mesh_infos.pkl        {frame: {'Rh' (3,), 'Th' (3,), 'poses' (72,), 'joints' (24,3), 'tpose_joints' (24,3)}}
    canonical_joints.pkl  {'joints': (24,3)}
    images/<frame>.png, masks/<frame>.png

written with ``pickle.dump`` of plain dicts of numpy arrays.  ``load_array_pickle`` reads them with an unpickler that
can build nothing but numpy arrays / scalars and builtin containers -- a dataset file cannot run code here.
Checkpoints (``.tar`` = torch.save of {'iter', 'network', 'optimizer'}, trainer.py:356-364) are read by
``train.load_checkpoint`` with ``weights_only=True``.

``Subject`` mirrors what the reference's Dataset classes hold per subject (core/data/human_nerf/train.py:40-135,
freeview.py:32-118); ``movement_frame`` / ``freeview_frame`` / ``train_frame`` assemble the dict one
``Dataset.__getitem__`` yields (train.py:481-755, freeview.py:172-280) -- by default WITHOUT the per-pixel numpy
pass: the frame carries its camera and posed bbox, and ``render.render_frames`` / ``ops.gen_rays`` produce rays,
near / far and ray_mask on the device.  ``host_rays=True`` runs the reference's numpy route (scene.py) instead.

cv2 is not importable here.  What the reference does through it:
  * cv2.Rodrigues (camera_util.py:39, 128): restated by the closed form (``rodrigues_cv``), PARITY UNPINNED
    against OpenCV's own numerics (agrees with the formula to rounding);
  * cv2.undistort (train.py:366-371) and cv2.resize INTER_LANCZOS4 / INTER_LINEAR (train.py:408-417): restated after
    OpenCV's definitions in imageproc.py (host, numpy) and csrc/hnrf_image.hip (device) -- fixed-point undistortion
    map and weights, 8-tap Lanczos with float weights and double accumulation, 2x2 box mean for the mask at scale
    1/2.  The two statements agree bit for bit (tests/test_gpu_image.py); against OpenCV's binaries they are PARITY
    UNPINNED, and a frame that went through either step says so: ``resize_parity='unpinned'`` (else 'exact').
"""
import io
import os
import pickle

import numpy as np

from . import imageproc, scene
from .config import cfg

# ------------------------------------------------------------------------------------------------ safe pickle reader
_NUMPY_GLOBALS = {
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
    ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
}
_BUILTIN_GLOBALS = {('collections', 'OrderedDict'), ('builtins', 'dict'), ('builtins', 'list'), ('builtins', 'tuple'),
                    ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex'),
                    ('builtins', 'bytearray')}


class _ArrayUnpickler(pickle.Unpickler):
    """Unpickler whose only reachable globals build numpy arrays / scalars / dtypes and builtin containers."""

    def find_class(self, module, name):
        if (module, name) in _NUMPY_GLOBALS:
            import numpy._core.multiarray as ma
            import numpy._core.numeric as nu
            return {'_reconstruct': ma._reconstruct, 'scalar': ma.scalar, 'ndarray': np.ndarray, 'dtype': np.dtype,
                    '_frombuffer': nu._frombuffer}[name]
        if (module, name) in _BUILTIN_GLOBALS:
            import builtins
            import collections
            return getattr(collections if module == 'collections' else builtins, name)
        raise pickle.UnpicklingError('refusing to load %s.%s: dataset pickles may only contain numpy arrays and '
                                     'builtin containers' % (module, name))


def load_array_pickle(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return _ArrayUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, 'rb') as f:
        return _ArrayUnpickler(f).load()


# ------------------------------------------------------------------------------------------------ camera helpers
def rodrigues_cv(rvec):
    """What cv2.Rodrigues(rvec)[0] computes: R = cos t I + (1 - cos t) r r^T + sin t [r]_x, r = rvec / t, t = |rvec|
    (identity for t = 0), float64.  Unlike body_util's variant there is no +1e-5 in the normalisation."""
    rvec = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(rvec)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    r = rvec / theta
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.cos(theta) * np.eye(3) + (1.0 - np.cos(theta)) * np.outer(r, r) + np.sin(theta) * K


def rotation_to_rvec(R):
    """What cv2.Rodrigues(R)[0][:, 0] computes for a rotation matrix: the axis-angle vector with angle in [0, pi]."""
    R = np.asarray(R, dtype=np.float64)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = 0.5 * np.linalg.norm(ax), np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    theta = np.arctan2(s, c)
    if s > 1e-9:
        return ax / (2.0 * s) * theta
    if c > 0:
        return np.zeros(3)
    # angle pi: axis from the diagonal of (R + I) / 2 = r r^T, signs from its largest column
    B = 0.5 * (R + np.eye(3))
    k = int(np.argmax(np.diag(B)))
    r = B[:, k] / np.sqrt(B[k, k])
    return r * theta


def rotate_bbox(bbox, rmtx):
    """tpose.py:102-124: axis-aligned box of the eight corners multiplied by ``rmtx`` from the right."""
    mn, mx = np.asarray(bbox['min_xyz']), np.asarray(bbox['max_xyz'])
    pts = np.array([[x, y, z] for x in (mn[0], mx[0]) for y in (mn[1], mx[1]) for z in (mn[2], mx[2])])
    rot = pts.dot(rmtx)
    return {'min_xyz': np.min(rot, axis=0), 'max_xyz': np.max(rot, axis=0)}


def apply_global_tfm_to_camera(E, Rh, Th):
    """Extrinsics in the frame of the body's root (camera_util.py:117-131)."""
    g = np.eye(4)
    rot = rodrigues_cv(Rh).T
    g[:3, :3] = rot
    g[:3, 3] = -rot.dot(np.asarray(Th, dtype=np.float64))
    return np.asarray(E).dot(np.linalg.inv(g))


def rotate_camera_by_frame_idx(extrinsics, frame_idx, trans=None, rotate_axis='y', period=196, inv_angle=False):
    """Orbit of the free-viewpoint renderer (camera_util.py:6-48, 82-114)."""
    angle = 2 * np.pi * (frame_idx / period)
    if inv_angle:
        angle = -angle
    inv_E = np.linalg.inv(extrinsics)
    camrot, campos = inv_E[:3, :3], inv_E[:3, 3].copy()
    if trans is not None:
        campos -= trans
    if camrot.T[1, 1] < 0.:
        angle = -angle
    vec = np.zeros(3)
    vec[{'x': 0, 'y': 1, 'z': 2}[rotate_axis]] = angle
    grot = rodrigues_cv(vec).astype('float32')
    rot_campos, rot_camrot = grot.dot(campos), grot.dot(camrot)
    if trans is not None:
        rot_campos += trans
    new_E = np.identity(4)
    new_E[:3, :3] = rot_camrot.T
    new_E[:3, 3] = -rot_camrot.T.dot(rot_campos)
    return new_E


def skeleton_to_bbox(skeleton, offset=None):
    """train.py:118-126."""
    offset = cfg.bbox_offset if offset is None else offset
    return {'min_xyz': np.min(skeleton, axis=0) - offset, 'max_xyz': np.max(skeleton, axis=0) + offset}


ROT_CAM_PARAMS = {'zju_mocap': {'rotate_axis': 'z', 'inv_angle': True},
                  'wild': {'rotate_axis': 'y', 'inv_angle': False}}            # freeview.py:26-29


# ------------------------------------------------------------------------------------------------ subject
class Subject:
    """Everything the reference's Dataset.__init__ loads for one prepared subject directory."""

    def __init__(self, dataset_path, skip=1, maxframes=-1, volume_size=None, require_images=False):
        self.dataset_path = dataset_path
        self.image_dir = os.path.join(dataset_path, 'images')
        joints = load_array_pickle(os.path.join(dataset_path, 'canonical_joints.pkl'))['joints']
        self.canonical_joints = joints.astype('float32')
        self.canonical_bbox = skeleton_to_bbox(self.canonical_joints)
        self.cameras = load_array_pickle(os.path.join(dataset_path, 'cameras.pkl'))
        self.mesh_infos = load_array_pickle(os.path.join(dataset_path, 'mesh_infos.pkl'))
        for info in self.mesh_infos.values():
            info['bbox'] = skeleton_to_bbox(info['joints'])
        g = int(volume_size or cfg.mweight_volume.volume_size)
        self.motion_weights_priors = scene.approx_gaussian_bone_volumes(
            self.canonical_joints, self.canonical_bbox['min_xyz'], self.canonical_bbox['max_xyz'], grid_size=g).astype('float32')
        self.cnl_gtfms = scene.get_canonical_global_tfms(self.canonical_joints)
        if os.path.isdir(self.image_dir):                   # the reference lists images/*.png (train.py:172-176)
            frames = sorted(os.path.splitext(f)[0] for f in os.listdir(self.image_dir) if f.endswith('.png'))
        elif require_images:
            raise FileNotFoundError(self.image_dir)
        else:
            frames = list(self.mesh_infos.keys())
        self.framelist_all = frames
        self.framelist = frames[::skip]
        if maxframes > 0:
            self.framelist = self.framelist[:maxframes]

    def __len__(self):
        return len(self.framelist)

    # -- pieces shared by every frame kind ------------------------------------------------------------------
    def _skeleton_entries(self, info):
        poses = info['poses'].astype('float32')
        dst_Rs, dst_Ts = scene.body_pose_to_body_RTs(poses, info['tpose_joints'].astype('float32'))
        mn = self.canonical_bbox['min_xyz'].astype('float32')
        mx = self.canonical_bbox['max_xyz'].astype('float32')
        scale = 2.0 / (mx - mn)
        assert np.all(scale >= 0)
        return {'dst_Rs': dst_Rs, 'dst_Ts': dst_Ts, 'cnl_gtfms': self.cnl_gtfms,
                'motion_weights_priors': self.motion_weights_priors,      # same array object every frame: the
                'cnl_bbox_min_xyz': mn, 'cnl_bbox_max_xyz': mx,           # renderer keeps it resident by identity
                'cnl_bbox_scale_xyz': scale, 'dst_posevec': poses[3:] + 1e-2}

    def _camera_entries(self, K, E, info, H, W, host_rays):
        """K already scaled to the image size; E with the global transform applied."""
        bbox = info['bbox']
        out = {'img_width': int(W), 'img_height': int(H)}
        if not host_rays:
            out.update(K=K.astype('float32'), E=E.astype('float32'),
                       ray_bbox_min_xyz=np.asarray(bbox['min_xyz'], 'float32'),
                       ray_bbox_max_xyz=np.asarray(bbox['max_xyz'], 'float32'))
            return out
        rays_o, rays_d = scene.get_rays_from_KRT(H, W, K, E[:3, :3], E[:3, 3])
        rays_o, rays_d = rays_o.reshape(-1, 3).copy(), rays_d.reshape(-1, 3).copy()
        near, far, ray_mask = scene.rays_intersect_3d_bbox(bbox, rays_o, rays_d)
        rays_o, rays_d = rays_o[ray_mask], rays_d[ray_mask]
        out.update(rays=np.stack([rays_o, rays_d, rays_d], axis=0).astype('float32'),      # [o, d, d_camera]: SURVEY 2.3
                   near=near[:, None].astype('float32'), far=far[:, None].astype('float32'), ray_mask=ray_mask)
        return out

    # -- frame kinds ------------------------------------------------------------------------------------------
    def movement_frame(self, idx, bgcolor=None, host_rays=False, image_size=None, load_image=False, device=None):
        """What the train dataset yields in ray_shoot_mode 'image' (movement / progress renders,
        train.py:481-755): the frame's own camera and pose.  ``image_size`` (H, W) is needed when the image is
        not loaded (camera-only rendering).  ``device`` (a GPU): the image goes through load_image_device -- the PNGs
        are decoded here, undistortion / composite / resize run on the device and ``raw_rgbs`` is a tensor there."""
        name = self.framelist[idx]
        info, cam = self.mesh_infos[name], self.cameras[name]
        bg = np.array(cfg.bgcolor if bgcolor is None else bgcolor, dtype='float32')
        out = {'frame_name': name, 'bgcolor': bg}
        img = None
        if load_image and device is not None and str(device).startswith('cuda') and not host_rays:
            img, _, flag = self.load_image_device(name, bg, device)
            H, W = int(img.shape[0]), int(img.shape[1])
            out['resize_parity'] = flag
        elif load_image:
            img, alpha, flag = self.load_image(name, bg)
            img = (img / 255.).astype('float32')
            H, W = img.shape[:2]
            out['resize_parity'] = flag
        else:
            H, W = image_size
        K = cam['intrinsics'][:3, :3].copy()
        K[:2] *= cfg.get('resize_img_scale', 1.0)
        E = apply_global_tfm_to_camera(cam['extrinsics'], info['Rh'].astype('float32'), info['Th'].astype('float32'))
        out.update(self._camera_entries(K, E, info, H, W, host_rays))
        out.update(self._skeleton_entries(info))
        if img is not None:
            out['raw_rgbs'] = img
            if host_rays:
                out['target_rgbs'] = img.reshape(-1, 3)[out['ray_mask']]
        return out

    def freeview_frame(self, idx, total_frames, train_frame_idx=0, src_type='zju_mocap', bgcolor=None, host_rays=False,
                       image_size=None):
        """freeview.py:172-280: the training frame ``train_frame_idx`` seen from a camera orbiting the subject."""
        name = self.framelist_all[train_frame_idx]
        info, cam = self.mesh_infos[name], self.cameras[name]
        Th = info['Th'].astype('float32')
        E = rotate_camera_by_frame_idx(cam['extrinsics'], idx, trans=Th, period=total_frames, **ROT_CAM_PARAMS[src_type])
        K = cam['intrinsics'].copy()
        K[:2] *= cfg.get('resize_img_scale', 1.0)
        E = apply_global_tfm_to_camera(E, info['Rh'].astype('float32'), Th)
        H, W = image_size
        out = {'frame_name': name, 'bgcolor': np.array([255., 255., 255.] if bgcolor is None else bgcolor, dtype='float32')}
        out.update(self._camera_entries(K, E, info, H, W, host_rays))
        out.update(self._skeleton_entries(info))
        return out

    TPOSE_RENDER_SIZE = 512                                   # tpose.py:22-25
    TPOSE_CAM_PARAMS = {'radius': 6.0, 'focal': 1250.}

    def tpose_frame(self, idx, total_frames, bgcolor=None, host_rays=False, image_size=None):
        """tpose.py:127-228 (run.py's 'tpose' mode): the canonical skeleton with zero pose, turned about the vertical
        axis by 2 pi idx / total_frames through its root rotation, seen from the fixed camera of tpose.py:65-84."""
        size = int(image_size or self.TPOSE_RENDER_SIZE)
        angle = 2 * np.pi / total_frames * idx
        add_rmtx = rodrigues_cv(np.array([0, -angle, 0], dtype='float32'))
        poses = np.zeros(72, dtype='float32')
        poses[:3] = rotation_to_rvec(add_rmtx.dot(rodrigues_cv(poses[:3])))
        info = {'poses': poses, 'tpose_joints': self.canonical_joints,
                'bbox': rotate_bbox(self.canonical_bbox, add_rmtx)}
        K, E = scene.tpose_camera(size, **self.TPOSE_CAM_PARAMS)
        out = {'bgcolor': np.array([255., 255., 255.] if bgcolor is None else bgcolor, dtype='float32')}
        out.update(self._camera_entries(K, E, info, size, size, host_rays))
        out.update(self._skeleton_entries(info))
        return out

    def train_frame(self, idx, bgcolor=None):
        """One training item (ray_shoot_mode 'patch', train.py:481-631): image + mask from disk, rays of
        cfg.patch.N_patches windows of cfg.patch.size^2 pixels drawn by the reference's sampler (global numpy generator),
        target patches and their masks.  Background: random colour per item when ``bgcolor`` is None (train.py:513-516)."""
        name = self.framelist[idx]
        info, cam = self.mesh_infos[name], self.cameras[name]
        bg = (np.random.rand(3) * 255.).astype('float32') if bgcolor is None else np.array(bgcolor, dtype='float32')
        img, alpha, flag = self.load_image(name, bg)
        img = (img / 255.).astype('float32')
        H, W = img.shape[:2]
        K = cam['intrinsics'][:3, :3].copy()
        K[:2] *= cfg.get('resize_img_scale', 1.0)
        E = apply_global_tfm_to_camera(cam['extrinsics'], info['Rh'].astype('float32'), info['Th'].astype('float32'))
        ent = self._camera_entries(K, E, info, H, W, host_rays=True)
        ray_mask = ent['ray_mask']
        sel, pinfo, div = scene.sample_patch_rays(ray_mask, alpha[:, :, 0] > 0., ray_mask.reshape(H, W),
                                                  int(cfg.patch.N_patches), int(cfg.patch.size), H, W,
                                                  subject_ratio=float(cfg.patch.sample_subject_ratio))
        targets = np.stack([img[y0:y1, x0:x1] for (x0, y0), (x1, y1) in zip(pinfo['xy_min'], pinfo['xy_max'])], axis=0)
        out = {'frame_name': name, 'bgcolor': bg, 'img_width': W, 'img_height': H, 'ray_mask': ray_mask,
               'rays': ent['rays'][:, sel], 'near': ent['near'][sel], 'far': ent['far'][sel],
               'patch_div_indices': div, 'patch_masks': pinfo['mask'], 'target_patches': targets,
               'target_rgbs': img.reshape(-1, 3)[ray_mask][sel], 'resize_parity': flag}
        out.update(self._skeleton_entries(info))
        return out

    # -- images -------------------------------------------------------------------------------------------------
    def image_size(self, frame_name):
        """(H, W) of a frame after cfg.resize_img_scale, from the PNG header (camera-only rendering needs no pixels)."""
        from PIL import Image
        with Image.open(os.path.join(self.image_dir, '%s.png' % frame_name)) as im:
            w, h = im.size
        scale = float(cfg.get('resize_img_scale', 1.0))
        return (h, w) if scale == 1.0 else imageproc.resized_size(h, w, scale)

    def decode_frame(self, frame_name):
        """The two PNGs of a frame as uint8 (H, W, 3) arrays, mask in 0..255 (train.py:352-364), and the camera's
        (K, D) when it has lens distortion to remove (train.py:366-371; an all-zero vector is the identity map)."""
        from PIL import Image
        orig = np.array(Image.open(os.path.join(self.image_dir, '%s.png' % frame_name)).convert('RGB'))
        alpha = np.array(Image.open(os.path.join(self.dataset_path, 'masks', '%s.png' % frame_name)).convert('RGB'))
        if alpha.max() == 1:
            alpha = alpha * 255
        cam = self.cameras.get(frame_name, {})
        lens = None
        if 'distortions' in cam and np.any(np.asarray(cam['distortions']) != 0):
            lens = (np.asarray(cam['intrinsics'], np.float64)[:3, :3], imageproc.distortion_vector(cam['distortions']))
        return orig, alpha.astype(np.uint8), lens

    def load_image(self, frame_name, bg_color):
        """train.py:351-417 (default branches): undistort image and mask, alpha-composite over ``bg_color`` (0..255),
        resize by cfg.resize_img_scale.  Returns img (H, W, 3) float in 0..255, alpha (H, W, 3) in 0..1, and
        'exact' | 'unpinned' (an OpenCV step ran through its restatement, imageproc.py)."""
        orig, alpha, lens = self.decode_frame(frame_name)
        scale = float(cfg.get('resize_img_scale', 1.0))
        img, a = imageproc.load_step(orig, alpha, bg_color, K=None if lens is None else lens[0],
                                     D=None if lens is None else lens[1], scale=scale)
        return img, a, 'exact' if lens is None and scale == 1.0 else 'unpinned'

    def load_image_device(self, frame_name, bg_color, device):
        """load_image on the GPU (hnrf_image.hip): the decoded PNGs are uploaded as bytes, undistortion, composite and
        resize run there.  Returns img / 255 as float32 (H, W, 3) on the device (what the frame dicts carry as
        ``raw_rgbs``), the resized mask's first channel (float32 (H, W)), and the parity flag."""
        import torch
        from . import ops
        orig, alpha, lens = self.decode_frame(frame_name)
        scale = float(cfg.get('resize_img_scale', 1.0))
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(device, non_blocking=True)
        with torch.cuda.device(device):
            o, a = up(orig), up(alpha)
            if lens is not None:
                o, a = ops.undistort_image(o, *lens), ops.undistort_image(a, *lens)
            bg = up(np.asarray(bg_color, dtype=np.float32))
            Hd, Wd = imageproc.resized_size(o.shape[0], o.shape[1], scale) if scale != 1.0 else o.shape[:2]
            img = ops.composite_windows(o, a, bg, [(0, 0)], Hd, Wd, scale=scale)[0]
            mask = ops.resize_mask(a, scale)
        return img, mask, 'exact' if lens is None and scale == 1.0 else 'unpinned'


class DeviceFrameCache:
    """Training items assembled on the device from per-frame data that stays resident in HBM.

    Everything Subject.train_frame computes except the random choices is a constant of the frame: the rays of the
    whole image, near / far, the bbox hit mask, the image and its alpha.  The numpy route recomputes all of it for
    every item (262 144 rays generated and slab-tested to keep 6 144: 230 ms per item on this container's cores --
    4 items/s per loader thread against a training step of 11 ms).  Here a frame's constants are built once --
    hnrf_gen_rays on the device (same hit mask, rays within 1 ulp of the numpy helpers:
    tests/test_gpu_parity.py::test_ray_generation_kernel), PNGs decoded once -- and kept on the GPU (11 MB per
    512x512 frame: a 650-frame ZJU subject is 7 GB of the 288); an item then costs the reference's patch draw on the
    host (scene.PatchSampler: same numpy generator calls, same result) and a few gathers / 32x32 crops on the device.

    ``train_batch(idx)`` returns what ``to_device(subject.train_frame(idx))`` returns, equal value for value given the
    same state of the global numpy generator (tests/test_gpu_parity.py::test_device_frame_cache_equals_host_route)."""

    def __init__(self, subject, device, max_bytes=64 << 30):
        self.subject, self.device, self.max_bytes = subject, device, int(max_bytes)
        self.entries, self.bytes = {}, 0
        import threading
        self._lock = threading.Lock()

    def _build(self, idx):
        import torch
        from . import ops
        subj, dev = self.subject, self.device
        name = subj.framelist[idx]
        info, cam = subj.mesh_infos[name], subj.cameras[name]
        scale = float(cfg.get('resize_img_scale', 1.0))
        orig, alpha, lens = subj.decode_frame(name)
        K = cam['intrinsics'][:3, :3].copy()
        K[:2] *= scale
        E = apply_global_tfm_to_camera(cam['extrinsics'], info['Rh'].astype('float32'), info['Th'].astype('float32'))
        bbox = info['bbox']
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True)
        with torch.cuda.device(dev):
            # the frame's two byte images stay resident at their SOURCE resolution, undistorted once (train.py:366-371);
            # composite + Lanczos run per item on the few windows it needs (hnrf_composite_windows)
            o, a = up(orig), up(alpha)
            if lens is not None:
                o, a = ops.undistort_image(o, *lens), ops.undistort_image(a, *lens)
            H, W = imageproc.resized_size(o.shape[0], o.shape[1], scale) if scale != 1.0 else tuple(o.shape[:2])
            subject_mask = (ops.resize_mask(a, scale) > 0).cpu().numpy()          # train.py:626 ``alpha[:, :, 0] > 0.``
            g = ops.gen_rays(K.astype('float32'), E.astype('float32'), np.asarray(bbox['min_xyz'], 'float32'),
                             np.asarray(bbox['max_xyz'], 'float32'), H, W, device=dev)
        ray_mask = g['ray_mask'].cpu().numpy()
        ent = {'name': name, 'H': H, 'W': W, 'scale': scale,
               'rays': g['rays'][:2].contiguous(), 'near': g['near'].contiguous(), 'far': g['far'].contiguous(),
               'ray_mask': g['ray_mask'], 'orig': o, 'alpha': a,
               'parity': 'exact' if lens is None and scale == 1.0 else 'unpinned',
               'sampler': scene.PatchSampler(ray_mask, subject_mask, ray_mask.reshape(H, W), H, W),
               'skeleton': {k: torch.as_tensor(np.ascontiguousarray(v)).to(dev)
                            for k, v in subj._skeleton_entries(info).items()}}
        ent['bytes'] = sum(v.numel() * v.element_size() for v in ent.values() if torch.is_tensor(v))
        return ent

    def entry(self, idx):
        with self._lock:
            ent = self.entries.get(idx)
        if ent is None:
            ent = self._build(idx)
            with self._lock:
                if idx not in self.entries and self.bytes + ent['bytes'] <= self.max_bytes:
                    self.entries[idx] = ent
                    self.bytes += ent['bytes']
        return ent

    def train_batch(self, idx, bgcolor=None, rng=None):
        """``rng`` None: the item's random choices come from the global numpy generator exactly like train_frame's (and
        the reference's); a numpy Generator: from that (scene.PatchSampler.draw)."""
        import torch
        ent, dev = self.entry(idx), self.device
        # the two random draws of an item, in train_frame's order: background colour, then the patches
        if bgcolor is not None:
            bg = np.array(bgcolor, dtype='float32')
        else:
            bg = ((np.random.rand(3) if rng is None else rng.random(3)) * 255.).astype('float32')
        sel, pinfo, div = ent['sampler'].draw(int(cfg.patch.N_patches), int(cfg.patch.size),
                                              subject_ratio=float(cfg.patch.sample_subject_ratio), rng=rng)
        # uploads through pinned memory: a pageable copy waits for everything queued on the stream (the training
        # step in flight) and would hold this loader thread for a whole step per tensor
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True)
        sel_d, masks_d, bg_d = up(sel.astype(np.int64)), up(pinfo['mask']), up(bg)
        # (positions of the kept pixels inside the stacked patches, found on the host: a boolean index on the device
        # would read the count back)
        flat_d = up(np.flatnonzero(pinfo['mask']))
        o, d = ent['rays'][0].index_select(0, sel_d), ent['rays'][1].index_select(0, sel_d)
        # float64 composite (+ Lanczos at cfg.resize_img_scale) of the windows' pixels only, rounded to float32 once at
        # the end like the numpy route (load_image, then ``img / 255`` as float32)
        from . import ops
        size = int(cfg.patch.size)
        targets = ops.composite_windows(ent['orig'], ent['alpha'], bg_d, pinfo['xy_min'], size, size, scale=ent['scale'])
        out = {'frame_name': ent['name'], 'bgcolor': bg_d, 'img_width': ent['W'], 'img_height': ent['H'],
               'ray_mask': ent['ray_mask'], 'rays': torch.stack([o, d, d], 0),
               'near': ent['near'].index_select(0, sel_d), 'far': ent['far'].index_select(0, sel_d),
               'patch_div_indices': torch.from_numpy(div), 'patch_masks': masks_d, 'target_patches': targets,
               'target_rgbs': targets.reshape(-1, 3).index_select(0, flat_d), 'resize_parity': ent['parity']}
        out.update(ent['skeleton'])
        return out


class FrameStream:
    """Endless iterator of training batches on the device: what DataLoader(train dataset, batch_size=1, shuffle=True,
    num_workers=cfg.num_workers) + cpu_data_to_gpu are to the reference's loop (create_dataset.py:73-88,
    trainer.py:193-199), for one process per GPU.  Every epoch is one seeded permutation of the frames -- the same on
    all ranks -- of which rank r takes elements r, r + world, ... (wrapping around so that every rank gets the same
    count: DistributedSampler's rule).  ``workers`` threads assemble frames ahead of the consumer (PNG decoding and the
    numpy ray / patch code release the GIL for most of their time), ``prefetch`` batches are kept ready; uploads go
    through pinned memory so that they overlap the step that is running."""

    def __init__(self, subject, rank=0, world=1, seed=0, device=None, prefetch=3, workers=2, bgcolor=None,
                 device_cache=None, exact_draws=None):
        import threading
        import torch
        if device_cache is None:                      # default: frames stay resident on a GPU (DeviceFrameCache)
            device_cache = device is not None and torch.device(device).type == 'cuda'
        self.cache = DeviceFrameCache(subject, device) if device_cache else None
        self.subject, self.rank, self.world, self.seed = subject, int(rank), int(world), int(seed)
        self.device, self.bgcolor = device, bgcolor
        # cfg.amd.exact_patch_draws: patch positions from the reference's own calls on the global numpy generator (4 ms
        # per item: choice(replace=False) shuffles every candidate pixel) instead of a Generator per item
        self.exact_draws = bool(cfg.get('amd', {}).get('exact_patch_draws', False)) if exact_draws is None else bool(exact_draws)
        self._done = {}
        self._cv = threading.Condition()
        self._next_put, self._next_get, self._stop = 0, 0, False
        self._order = self._indices()
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, workers))]
        self._slots = threading.Semaphore(max(1, prefetch))
        self._err = None
        for t in self._threads:
            t.start()

    def _indices(self):
        epoch = 0
        while True:
            n = len(self.subject)
            perm = np.random.RandomState(self.seed + epoch).permutation(n)
            per_rank = -(-n // self.world)
            padded = np.resize(perm, per_rank * self.world)
            for i in padded[self.rank::self.world]:
                yield int(i)
            epoch += 1

    def _work(self):
        while True:
            self._slots.acquire()
            with self._cv:
                if self._stop:
                    return
                ticket, idx = self._next_put, next(self._order)
                self._next_put += 1
            try:
                if self.cache is not None:
                    import torch
                    # the item's own generator, keyed by (seed, rank, ticket): the same items whatever the threads do
                    rng = None if self.exact_draws else np.random.default_rng([self.seed, self.rank, ticket])
                    with torch.cuda.device(self.device):
                        item = self.cache.train_batch(idx, bgcolor=self.bgcolor, rng=rng)
                else:
                    item = self.subject.train_frame(idx, bgcolor=self.bgcolor)
            except Exception as e:                                   # surfaced by __next__
                item = e
            with self._cv:
                self._done[ticket] = item
                self._cv.notify_all()

    def __iter__(self):
        return self

    def __next__(self):
        with self._cv:
            while self._next_get not in self._done:
                self._cv.wait()
            item = self._done.pop(self._next_get)
            self._next_get += 1
        self._slots.release()
        if isinstance(item, Exception):
            self.close()
            raise item
        if self.cache is not None:
            exclude = ('frame_name', 'img_width', 'img_height', 'resize_parity')
            return {k: v for k, v in item.items() if k not in exclude}
        return to_device(item, self.device, pinned=True) if self.device is not None else item

    def close(self):
        with self._cv:
            self._stop = True
        for _ in self._threads:
            self._slots.release()


def to_device(batch, device, exclude=('frame_name', 'img_width', 'img_height', 'resize_parity'), pinned=False,
              host_keys=('patch_div_indices',)):
    """cpu_data_to_gpu (train_util.py:7-25): tensors of everything but the excluded keys.  ``host_keys`` become tensors
    but stay on the host: the patch boundaries are read as Python ints by the loss (trainer.py:28-37) -- on the device
    every one of them would be a synchronisation."""
    import torch
    out = {}
    for k, v in batch.items():
        if k in exclude or isinstance(v, str):
            continue
        if k in host_keys:
            out[k] = torch.as_tensor(v)
            continue
        t = torch.as_tensor(np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v)
        if pinned and torch.device(device).type == 'cuda' and t.numel() > 0:
            t = t.pin_memory()
        out[k] = t.to(device, non_blocking=pinned)
    return out
