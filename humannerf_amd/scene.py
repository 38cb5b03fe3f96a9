"""Per-frame inputs of the renderer: skeleton kinematics, weight-volume priors,
camera rays and ray/bbox intersection (host side, numpy).

These are the producers of the dict consumed by ``Network.forward`` -- the
"next" row of SURVEY.md section 8(f) rank 1.  Each function states the reference
routine whose *result* it reproduces; the code is vectorised and written for
this repo (no per-bone python loops over voxels, no cv2).

Also provides ``synthetic_frame`` -- the synthetic T-pose-skeleton scene of
SURVEY.md Appendix A.2 used by bench.py, smoke() and the golden fixtures (no
dataset, SMPL model or checkpoint is available offline).
"""
import numpy as np

# SMPL kinematic tree (reference: core/utils/body_util.py:32-35).
SMPL_PARENT = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13,
                        14, 16, 17, 18, 19, 20, 21], dtype=np.int64)
TORSO_JOINTS = (0, 3, 6, 9, 13, 14)        # body_util.py:37-42
HEAD_JOINT = 15
BONE_STDS = np.array([0.03, 0.06, 0.03])   # body_util.py:43-45
HEAD_STDS = np.array([0.06, 0.06, 0.06])
JOINT_STDS = np.array([0.02, 0.02, 0.02])

# Synthetic 24-joint T-pose (metres), SURVEY.md Appendix A.2.
TPOSE_JOINTS = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00],
    [0.00, 0.11, -0.02], [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00],
    [0.00, 0.25, 0.00], [0.09, -0.87, -0.03], [-0.09, -0.87, -0.03],
    [0.00, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.00, 0.51, -0.01], [0.08, 0.42, 0.00], [-0.08, 0.42, 0.00],
    [0.00, 0.60, 0.03], [0.18, 0.44, 0.00], [-0.18, 0.44, 0.00],
    [0.43, 0.44, -0.01], [-0.43, 0.44, -0.01], [0.68, 0.44, 0.00],
    [-0.68, 0.44, 0.00], [0.77, 0.43, 0.00], [-0.77, 0.43, 0.00],
], dtype=np.float32)


def _skew(v):
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=np.float64)


def rodrigues(rvec):
    """Axis-angle -> 3x3, with the reference's ``norm + 1e-5`` axis
    normalisation (body_util.py:200-219)."""
    rvec = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(rvec)
    r = rvec / (theta + 1e-5)
    return (np.cos(theta) * np.eye(3) + np.sin(theta) * _skew(r)
            + (1.0 - np.cos(theta)) * np.outer(r, r))


def body_pose_to_body_RTs(jangles, tpose_joints):
    """Local joint rotations / parent-relative offsets.
    Result of body_util.py:222-248: Rs (J,3,3) f32, Ts (J,3) f32."""
    jangles = np.asarray(jangles).reshape(-1, 3)
    J = jangles.shape[0]
    Rs = np.stack([rodrigues(jangles[i]) for i in range(J)]).astype(np.float32)
    Ts = tpose_joints.astype(np.float32).copy()
    Ts[1:] = tpose_joints[1:] - tpose_joints[SMPL_PARENT[1:]]
    return Rs, Ts


def get_canonical_global_tfms(canonical_joints):
    """Global 4x4 transforms of the rest pose: pure translations accumulated
    down the tree (body_util.py:251-271)."""
    J = canonical_joints.shape[0]
    g = np.zeros((J, 4, 4), dtype=np.float32)
    g[:, 3, 3] = 1.0
    g[:, :3, :3] = np.eye(3, dtype=np.float32)
    local = canonical_joints.astype(np.float32).copy()
    local[1:] = canonical_joints[1:] - canonical_joints[SMPL_PARENT[1:]]
    for i in range(J):
        # float32 products of [I|t] matrices, parent first (same association).
        loc = np.eye(4, dtype=np.float32)
        loc[:3, 3] = local[i]
        g[i] = loc if i == 0 else g[SMPL_PARENT[i]].dot(loc)
    return g


def _align_y_to(direction):
    """Rotation taking +y onto ``direction`` (body_util.py:83-116)."""
    v1 = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    v2 = direction.astype(np.float32)
    v1 = v1 / np.clip(np.linalg.norm(v1), 1e-5, None)
    v2 = v2 / np.clip(np.linalg.norm(v2), 1e-5, None)
    n = np.cross(v1, v2)
    c = float(v1.dot(v2))
    K = _skew(n).astype(np.float32)
    return (np.eye(3) + K + K.dot(K) * (1.0 / (1.0 + c))).astype(np.float32)


def _gaussian_volume(grid_xyz, center, inv_std, rot):
    """exp(-d^T R S S R^T d) on the grid (body_util.py:136-178)."""
    S = np.diag(inv_std.astype(np.float32))
    sigma = rot.dot(S).dot(S).dot(rot.T)
    d = grid_xyz - np.asarray(center)
    q = np.einsum('...i,ij,...j->...', d, sigma, d)
    return np.exp(-q)


def approx_gaussian_bone_volumes(tpose_joints, bbox_min_xyz, bbox_max_xyz,
                                 grid_size=32):
    """Priors for the motion-weight volume: (J+1, G, G, G), background last,
    indexed [z][y][x] (body_util.py:274-348)."""
    tpose_joints = tpose_joints.astype(np.float32)
    J = tpose_joints.shape[0]
    ax = [np.linspace(bbox_min_xyz[i], bbox_max_xyz[i], grid_size) for i in range(3)]
    zg, yg, xg = np.meshgrid(ax[2], ax[1], ax[0], indexing='ij')
    grid = np.stack([xg, yg, zg], axis=-1)

    vols = []
    for j in range(J):
        children = np.nonzero(SMPL_PARENT == j)[0]
        if len(children) > 0:
            vol = np.zeros((grid_size,) * 3, dtype=np.float32)
            for c in children:
                inv_std = 1.0 / (BONE_STDS * 2.0)
                inv_std = inv_std.astype(np.float32)
                if j in TORSO_JOINTS:
                    inv_std[0] *= 1 / 1.5
                    inv_std[2] *= 1 / 1.5
                start, end = tpose_joints[j], tpose_joints[c]
                rot = _align_y_to(end - start)
                vol = vol + _gaussian_volume(grid, (start + end) / 2.0, inv_std, rot)
        else:
            stds = HEAD_STDS if j == HEAD_JOINT else JOINT_STDS
            inv_std = (1.0 / (stds * 2.0)).astype(np.float32)
            vol = _gaussian_volume(grid, tpose_joints[j], inv_std,
                                   np.eye(3, dtype=np.float32))
        vols.append(vol)
    vols = np.stack(vols, axis=0)
    bg = 1.0 - np.sum(vols, axis=0, keepdims=True).clip(min=0.0, max=1.0)
    vols = np.concatenate([vols, bg], axis=0)
    return vols / np.sum(vols, axis=0, keepdims=True).clip(min=0.001)


def get_camrot(campos, lookat=None, inv_camera=False):
    """Rows = right, up, forward (camera_util.py:50-79)."""
    lookat = np.zeros(3, dtype=np.float32) if lookat is None else np.asarray(lookat)
    up = np.array([0.0, -1.0 if inv_camera else 1.0, 0.0], dtype=np.float32)
    fwd = lookat - campos
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(up, fwd)
    right = right / np.linalg.norm(right)
    up = np.cross(fwd, right)
    up = up / np.linalg.norm(up)
    return np.array([right, up, fwd], dtype=np.float32)


def get_rays_from_KRT(H, W, K, R, T):
    """Pixel rays in world space, direction NOT normalised
    (camera_util.py:132-159)."""
    rays_o = -np.dot(R.T, T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32),
                       np.arange(H, dtype=np.float32), indexing='xy')
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pix_cam = np.dot(xy1, np.linalg.inv(K).T)
    pix_world = np.dot(pix_cam - T.ravel(), R)
    rays_d = pix_world - rays_o[None, None]
    return np.broadcast_to(rays_o, rays_d.shape), rays_d


def rays_intersect_3d_bbox(bounds, ray_o, ray_d):
    """Slab test against the bbox padded by 1 cm; keeps rays that hit exactly
    two faces.  Returns near, far (valid rays only) and the hit mask
    (camera_util.py:162-208).  ``ray_d`` is clamped in place like the
    reference does."""
    if isinstance(bounds, dict):
        bounds = np.stack([bounds['min_xyz'], bounds['max_xyz']], axis=0)
    assert bounds.shape == (2, 3)
    bounds = bounds + np.array([-0.01, 0.01])[:, None]
    ray_d[np.abs(ray_d) < 1e-5] = 1e-5
    t = ((bounds[None] - ray_o[:, None]) / ray_d[:, None]).reshape(-1, 6)
    p = t[..., None] * ray_d[:, None] + ray_o[:, None]
    lo, hi = bounds[0] - 1e-6, bounds[1] + 1e-6
    inside = np.all((p >= lo) & (p <= hi), axis=-1)
    hit = inside.sum(-1) == 2
    pts = p[hit][inside[hit]].reshape(-1, 2, 3)
    o, d = ray_o[hit], ray_d[hit]
    nrm = np.linalg.norm(d, axis=1)
    d0 = np.linalg.norm(pts[:, 0] - o, axis=1) / nrm
    d1 = np.linalg.norm(pts[:, 1] - o, axis=1) / nrm
    return np.minimum(d0, d1), np.maximum(d0, d1), hit


def sample_patch_rays(ray_mask, subject_mask, bbox_mask, n_patch, patch_size, H, W, subject_ratio=0.8):
    """Training-time ray selection (core/data/human_nerf/train.py:236-335): ``n_patch`` square windows whose
    centres are drawn on the subject (probability ``subject_ratio``) or on the rest of the projected bbox; of each
    window the pixels whose ray crosses the bbox are kept.  Draws from the GLOBAL numpy generator in the
    reference's order (one ``rand`` and one ``choice`` per patch), so a seeded run picks the same patches.

    ray_mask (H*W,) bool; subject_mask, bbox_mask (H, W) bool.  Returns
    select_inds (indices into the compacted ray arrays), patch_info {'mask' (n,ps,ps), 'xy_min', 'xy_max' (n,2)},
    patch_div_indices (n+1,)."""
    assert ray_mask.dtype == bool and subject_mask.dtype == bool and bbox_mask.dtype == bool and ray_mask.ndim == 1
    outside_subject = bbox_mask & ~subject_mask
    compact_index = np.cumsum(ray_mask) - 1                    # pixel -> position among the kept rays
    inds, masks, xy_min, xy_max, div = [], [], [], [], [0]
    for _ in range(n_patch):
        cand = subject_mask if np.random.rand(1)[0] < subject_ratio else outside_subject
        ys, xs = np.where(cand)
        pick = np.random.choice(ys.shape[0], size=[1], replace=False)[0]
        half = patch_size // 2
        x0 = np.clip(xs[pick] - half, 0, W - patch_size)
        y0 = np.clip(ys[pick] - half, 0, H - patch_size)
        window = np.zeros((H, W), dtype=bool)
        window[y0:y0 + patch_size, x0:x0 + patch_size] = True
        hit = window.reshape(-1) & ray_mask
        inds.append(compact_index[np.where(hit)])
        masks.append(hit.reshape(H, W)[y0:y0 + patch_size, x0:x0 + patch_size])
        xy_min.append(np.array([x0, y0]))
        xy_max.append(np.array([x0 + patch_size, y0 + patch_size]))
        div.append(div[-1] + len(inds[-1]))
    return (np.concatenate(inds, axis=0),
            {'mask': np.stack(masks, axis=0), 'xy_min': np.stack(xy_min, axis=0), 'xy_max': np.stack(xy_max, axis=0)},
            np.array(div))


class PatchSampler:
    """sample_patch_rays for ONE frame drawn many times: the candidate pixel lists, the pixel -> ray index map and the
    2-D hit mask are built once, a draw touches only the n_patch windows.  Same calls on the global numpy generator in
    the same order (one ``rand``, one ``choice(n, size=[1], replace=False)`` per patch), same outputs as
    sample_patch_rays -- tests/test_host_cpu.py checks equality draw for draw."""

    def __init__(self, ray_mask, subject_mask, bbox_mask, H, W):
        assert ray_mask.dtype == bool and subject_mask.dtype == bool and bbox_mask.dtype == bool and ray_mask.ndim == 1
        self.H, self.W = H, W
        self.hit = ray_mask.reshape(H, W)
        self.compact_index = (np.cumsum(ray_mask) - 1).reshape(H, W)
        self.on = np.where(subject_mask)
        self.off = np.where(bbox_mask & ~subject_mask)

    def draw(self, n_patch, patch_size, subject_ratio=0.8, rng=None):
        """``rng`` None: the reference's calls on the GLOBAL numpy generator, draw for draw -- including
        ``choice(n, size=[1], replace=False)``, which shuffles all n candidate pixels to return one (0.6 ms per patch at
        512x512, under the GIL: 4 ms per item, more than a third of a training step).  ``rng`` = a numpy Generator: the same
        distribution from two numbers per patch (what dataset.FrameStream uses unless cfg.amd.exact_patch_draws)."""
        H, W = self.H, self.W
        inds, masks, xy_min, xy_max, div = [], [], [], [], [0]
        for _ in range(n_patch):
            if rng is None:
                ys, xs = self.on if np.random.rand(1)[0] < subject_ratio else self.off
                pick = np.random.choice(ys.shape[0], size=[1], replace=False)[0]
            else:
                ys, xs = self.on if rng.random() < subject_ratio else self.off
                pick = int(rng.integers(ys.shape[0]))
            half = patch_size // 2
            x0 = np.clip(xs[pick] - half, 0, W - patch_size)
            y0 = np.clip(ys[pick] - half, 0, H - patch_size)
            hit = self.hit[y0:y0 + patch_size, x0:x0 + patch_size]
            inds.append(self.compact_index[y0:y0 + patch_size, x0:x0 + patch_size][hit])     # row-major, like np.where
            masks.append(hit.copy())
            xy_min.append(np.array([x0, y0]))
            xy_max.append(np.array([x0 + patch_size, y0 + patch_size]))
            div.append(div[-1] + len(inds[-1]))
        return (np.concatenate(inds, axis=0),
                {'mask': np.stack(masks, axis=0), 'xy_min': np.stack(xy_min, axis=0), 'xy_max': np.stack(xy_max, axis=0)},
                np.array(div))


def tpose_camera(img_size, radius=6.0, focal=1250.0):
    """Orbit camera of the reference's T-pose renderer
    (core/data/human_nerf/tpose.py:65-84)."""
    campos = np.array([0.0, -0.25, radius], dtype=np.float32)
    camrot = get_camrot(campos, lookat=np.array([0, -0.25, 0.0]), inv_camera=True)
    E = np.eye(4, dtype=np.float32)
    E[:3, :3] = camrot
    E[:3, 3] = -camrot.dot(campos)
    K = np.eye(3, dtype=np.float32)
    K[0, 0] = K[1, 1] = focal
    K[:2, 2] = img_size / 2.0
    return K, E


def synthetic_frame(H=512, W=512, pose_seed=0, pose_scale=0.2, focal_at_512=1700.0,
                    bbox_offset=0.3, volume_size=32, bgcolor=(0.0, 0.0, 0.0),
                    ray_stride=1, camera_only=False):
    """The synthetic frame of SURVEY.md Appendix A.2 / section 8(d).

    Returns the numpy dict a reference dataset would yield for one frame
    (keys as consumed by Network.forward).  ``focal_at_512=1700`` makes every
    pixel ray hit the canonical bbox so that R == H*W exactly; 1250 is the
    T-pose renderer's framing (about 88 % of the pixels hit).
    ``ray_stride`` sub-samples the pixel grid (golden fixtures use few rays).
    ``camera_only``: leave out rays / near / far / ray_mask and return the camera (K, E) instead, for
    the device-side ray generator.
    """
    J = TPOSE_JOINTS
    mn, mx = J.min(0) - bbox_offset, J.max(0) + bbox_offset
    priors = approx_gaussian_bone_volumes(J, mn, mx, volume_size).astype(np.float32)
    poses = np.random.RandomState(pose_seed).randn(72) * pose_scale
    poses[:3] = 0.0
    dst_Rs, dst_Ts = body_pose_to_body_RTs(poses, J)
    cnl_gtfms = get_canonical_global_tfms(J)
    dst_posevec = (poses[3:] + 1e-2).astype(np.float32)

    K, E = tpose_camera(np.array([W, H], dtype=np.float32), 6.0,
                        focal_at_512 * H / 512.0)
    common = {
        'dst_Rs': dst_Rs, 'dst_Ts': dst_Ts, 'cnl_gtfms': cnl_gtfms,
        'motion_weights_priors': priors,
        'dst_posevec': dst_posevec,
        'cnl_bbox_min_xyz': mn.astype(np.float32),
        'cnl_bbox_max_xyz': mx.astype(np.float32),
        'cnl_bbox_scale_xyz': (2.0 / (mx - mn)).astype(np.float32),
        'bgcolor': np.array(bgcolor, dtype=np.float32),
        'img_width': W, 'img_height': H,
    }
    if camera_only:
        assert ray_stride == 1
        return dict(common, K=K, E=E)
    rays_o, rays_d = get_rays_from_KRT(H, W, K, E[:3, :3], E[:3, 3])
    rays_o = rays_o[::ray_stride, ::ray_stride].reshape(-1, 3).astype(np.float32)
    rays_d = rays_d[::ray_stride, ::ray_stride].reshape(-1, 3).astype(np.float32)
    rays_d = rays_d.copy()
    near, far, hit = rays_intersect_3d_bbox(np.stack([mn, mx]), rays_o, rays_d)
    rays_o, rays_d = rays_o[hit], rays_d[hit]
    return {
        'rays': np.stack([rays_o, rays_d, rays_d], axis=0).astype(np.float32),
        'near': near[:, None].astype(np.float32),
        'far': far[:, None].astype(np.float32),
        'ray_mask': hit,
        **common,
    }


# ------------------------------------------------------------------------------------------------ synthetic subject
# A subject DIRECTORY in the reference's on-disk layout (tools/prepare_zju_mocap/prepare_dataset.py:172-221) made of
# analytic content: no dataset is obtainable offline, and with closed-form images the image pre-processing
# (undistortion, resize) can be checked against the function it should recover.
ZJU_LIKE_DISTORTION = (-0.27, 0.11, -3e-4, 6e-4, -0.02)      # k1 k2 p1 p2 k3: the size ZJU-MoCap's calibration reports


def analytic_image(x, y, size, seed=0):
    """Smooth RGB pattern (0..255 float64, shape x.shape + (3,)) of the undistorted pixel coordinates; wavelengths of
    size/6 pixels and longer, so bilinear / Lanczos interpolation errors stay far below one grey level."""
    rs = np.random.RandomState(1000 + seed)
    out = []
    for _ in range(3):
        a, b, c, d = rs.uniform(3.0, 6.0, 4) * 2 * np.pi / size
        p, q = rs.uniform(0, 2 * np.pi, 2)
        out.append(127.5 + 60.0 * np.sin(a * x + b * y + p) + 55.0 * np.cos(c * x - d * y + q))
    return np.stack(out, axis=-1)


def analytic_mask(x, y, size):
    """Soft-edged ellipse (0..255 float64): 255 inside, a ramp of ~size/40 pixels to 0."""
    r = np.sqrt(((y - size / 2) / (size * 0.4)) ** 2 + ((x - size / 2) / (size * 0.2)) ** 2)
    return np.clip((1.0 - r) * 16.0, 0.0, 1.0) * 255.0


def distort_pixels(x, y, K, D):
    """Brown-Conrady forward model: undistorted pixel -> where the lens images it (the map cv2.undistort samples)."""
    k1, k2, p1, p2, k3 = [float(v) for v in D]
    xn, yn = (x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1]
    r2 = xn * xn + yn * yn
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = xn * kr + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * kr + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def undistort_pixels(u, v, K, D, iters=30):
    """Inverse of distort_pixels by fixed-point iteration (converges for the mild lenses used here)."""
    k1, k2, p1, p2, k3 = [float(t) for t in D]
    xd, yd = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
    xn, yn = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = xn * xn + yn * yn
        kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        dx = 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
        dy = p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
        xn, yn = (xd - dx) / kr, (yd - dy) / kr
    return K[0, 0] * xn + K[0, 2], K[1, 1] * yn + K[1, 2]


def write_synthetic_subject(out_dir, n_frames=4, size=512, distortions=None, seed=0, radius=4.0, binary_mask=False):
    """Write cameras.pkl / mesh_infos.pkl / canonical_joints.pkl / images / masks for ``n_frames`` frames of
    ``size`` x ``size`` pixels.  With ``distortions`` (k1 k2 p1 p2 k3) the PNGs hold what such a lens would record of
    analytic_image / analytic_mask, i.e. undistorting them must give those functions back on the pixel grid.
    Returns the frame names."""
    import os
    import pickle
    from PIL import Image
    os.makedirs(os.path.join(out_dir, 'images'), exist_ok=True)
    os.makedirs(os.path.join(out_dir, 'masks'), exist_ok=True)
    rs = np.random.RandomState(seed)
    J = TPOSE_JOINTS.astype(np.float64)
    D = np.zeros(5) if distortions is None else np.asarray(distortions, dtype=np.float64)
    cams, infos, names = {}, {}, []
    vv, uu = np.mgrid[0:size, 0:size].astype(np.float64)
    for n in range(n_frames):
        name = 'frame_%06d' % n
        names.append(name)
        K, E = tpose_camera(np.array([size, size], dtype=np.float32), radius, 1250.0 * size / 512.0)
        K = K.astype(np.float64)
        cams[name] = {'intrinsics': K, 'extrinsics': E.astype(np.float64), 'distortions': D.copy()}
        infos[name] = {'Rh': np.zeros(3), 'Th': np.zeros(3), 'poses': rs.randn(72) * 0.1, 'joints': J, 'tpose_joints': J}
        x, y = (uu, vv) if not np.any(D) else undistort_pixels(uu, vv, K, D)
        img = np.clip(np.rint(analytic_image(x, y, size, seed=n)), 0, 255).astype(np.uint8)
        m = analytic_mask(x, y, size)
        m = np.where(m > 127.5, 255.0, 0.0) if binary_mask else np.rint(m)
        Image.fromarray(img).save(os.path.join(out_dir, 'images', name + '.png'))
        Image.fromarray(np.stack([m.astype(np.uint8)] * 3, -1)).save(os.path.join(out_dir, 'masks', name + '.png'))
    for fname, obj in (('cameras.pkl', cams), ('mesh_infos.pkl', infos), ('canonical_joints.pkl', {'joints': J})):
        with open(os.path.join(out_dir, fname), 'wb') as f:
            pickle.dump(obj, f)
    return names
