"""Generate the golden fixtures under tests/golden/ by running the REFERENCE
(imported read-only from /root/reference, CPU) on seeded weights and the
synthetic frame of humannerf_amd/scene.py.

Run in the build container only (the reference does not exist on the GPU box):

    python oracle/make_golden.py            # writes tests/golden/*.npz + meta.json

Harness recipe: SURVEY.md section 8(c).  Nothing from the reference is copied; only
its numerical outputs are stored.  Also cross-checks this repo's numpy scene
helpers against the reference's own helpers, and records the reference's CPU
rays/s (indicative baseline).
"""
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(REPO, 'tests', 'golden')

ORACLE_YAML = """\
task: 'zju_mocap'
subject: 'p387'
experiment: 'oracle_cpu'
primary_gpus: ['cpu']
secondary_gpus: ['cpu']
bgcolor: [0., 0., 0.]
resize_img_scale: 0.5
"""


def import_reference():
    sys.dont_write_bytecode = True
    os.makedirs('/tmp/hnrf_oracle', exist_ok=True)
    ypath = '/tmp/hnrf_oracle/oracle.yaml'
    with open(ypath, 'w') as f:
        f.write(ORACLE_YAML)
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.path.insert(1, REPO)
    sys.argv = ['make_golden', '--cfg', ypath]
    for name in ['cv2', 'torchvision', 'torchvision.models', 'torchvision.transforms']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision.transforms'].Compose = lambda *a, **k: None
    sys.modules['torchvision.transforms'].Normalize = lambda *a, **k: None
    from configs import cfg
    from core.nets import create_network
    return cfg, create_network


def main():
    import numpy as np
    import torch
    torch.set_num_threads(8)
    cfg, create_network = import_reference()
    from humannerf_amd import scene
    from humannerf_amd.seeded import seeded_state, with_density

    # ---- cross-check the scene helpers against the reference's numpy helpers
    from core.utils import body_util as rb, camera_util as rc
    J = scene.TPOSE_JOINTS
    mn, mx = J.min(0) - 0.3, J.max(0) + 0.3
    poses = np.random.RandomState(0).randn(72) * 0.2
    poses[:3] = 0
    a = rb.body_pose_to_body_RTs(poses, J)
    b = scene.body_pose_to_body_RTs(poses, J)
    assert np.allclose(a[0], b[0], atol=1e-6) and np.allclose(a[1], b[1], atol=0)
    assert np.allclose(rb.get_canonical_global_tfms(J), scene.get_canonical_global_tfms(J), atol=1e-7)
    pa = rb.approx_gaussian_bone_volumes(J, mn, mx, 32)
    pb = scene.approx_gaussian_bone_volumes(J, mn, mx, 32)
    print('priors max|diff|', np.abs(pa - pb).max())
    assert np.allclose(pa, pb, atol=2e-6)
    K, E = scene.tpose_camera(np.array([64., 64.], dtype=np.float32), 6.0, 1250 * 64 / 512)
    ro, rd = rc.get_rays_from_KRT(64, 64, K, E[:3, :3], E[:3, 3])
    so, sd = scene.get_rays_from_KRT(64, 64, K, E[:3, :3], E[:3, 3])
    assert np.array_equal(ro, so) and np.array_equal(rd, sd)
    n1 = rc.rays_intersect_3d_bbox(np.stack([mn, mx]), ro.reshape(-1, 3).copy(), rd.reshape(-1, 3).copy())
    n2 = scene.rays_intersect_3d_bbox(np.stack([mn, mx]), so.reshape(-1, 3).copy(), sd.reshape(-1, 3).copy())
    assert all(np.array_equal(x, y) for x, y in zip(n1, n2))
    assert np.allclose(rc.get_camrot(np.array([0, -.25, 6.], dtype=np.float32), np.array([0, -.25, 0.]), True),
                       scene.get_camrot(np.array([0, -.25, 6.], dtype=np.float32), np.array([0, -.25, 0.]), True))
    print('scene helpers agree with the reference helpers')

    # ---- reference network with seeded parameters
    net = create_network()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    state = seeded_state(shapes, seed=0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    net.eval()
    Net = type(net)

    captured = {}
    orig_smf = Net._sample_motion_fields
    orig_r2o = Net._raw2outputs

    def smf(**kw):
        r = orig_smf(**kw)
        captured.setdefault('x_skel', []).append(r['x_skel'].detach())
        captured.setdefault('mask', []).append(r['fg_likelihood_mask'].detach()[..., 0])
        return r

    def r2o(raw, raw_mask, z_vals, rays_d, xyz, bgcolor=None):
        captured.setdefault('raw', []).append(raw.detach())
        captured.setdefault('z_vals', []).append(z_vals.detach())
        return orig_r2o(raw, raw_mask, z_vals, rays_d, xyz, bgcolor)

    Net._sample_motion_fields = staticmethod(smf)
    Net._raw2outputs = staticmethod(r2o)

    def frame_tensors(fr):
        keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors',
                'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
        d = {k: torch.from_numpy(np.ascontiguousarray(fr[k])) for k in keys}
        d['head_id'] = torch.tensor(-1)
        return d

    base_state = state

    def run_case(name, iter_val, S, perturb=0.0, ignore_nr=False, with_t_rand=False, keep_rays=64, frame=None,
                 density=None, pose_kick_in=None):
        """One fixture.  ``frame``: overrides of the golden frame's synthetic_frame arguments; ``density``:
        (bias_delta, gain) for seeded.with_density; ``pose_kick_in``: cfg.pose_decoder.kick_in_iter for this case
        (the wild configs set it to 20000)."""
        fa = dict(FRAME, **(frame or {}))
        fr = scene.synthetic_frame(**fa)
        st = base_state if density is None else with_density(base_state, *density)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        cfg.N_samples = S
        cfg.perturb = perturb
        cfg.ignore_non_rigid_motions = ignore_nr
        had = cfg.pose_decoder.get('kick_in_iter', None)
        if pose_kick_in is not None:
            cfg.pose_decoder.kick_in_iter = pose_kick_in
        captured.clear()
        real_rand = torch.rand
        t_rand = None
        if with_t_rand:
            t_rand = np.random.RandomState(7).rand(fr['rays'].shape[1], S).astype(np.float32)
            torch.rand = lambda *a, **k: torch.from_numpy(t_rand)
        try:
            with torch.no_grad():
                out = net(**frame_tensors(fr), iter_val=iter_val)
        finally:
            torch.rand = real_rand
            if pose_kick_in is not None:
                if had is None:
                    cfg.pose_decoder.pop('kick_in_iter')
                else:
                    cfg.pose_decoder.kick_in_iter = had
        save = {}
        per_sample = ('weights_on_rays', 'xyz_on_rays', 'rgb_on_rays', 'backward_motion_weights', 'offsets')
        for k, v in out.items():
            v = v.numpy()
            save[k] = v[:keep_rays] if k in per_sample else v
        for k, v in captured.items():
            save['_' + k] = torch.cat(v, 0).numpy()[:keep_rays]
        if t_rand is not None:
            save['t_rand'] = t_rand
        save['_vol_slice'] = net.motion_weights_vol.detach().numpy()[:, 12:20:3, 8:24:5, 8:24:5]
        meta = dict(iter_val=float(iter_val), N_samples=S, perturb=perturb, ignore_non_rigid_motions=ignore_nr,
                    n_rays=int(out['rgb'].shape[0]), keep_rays=keep_rays, frame=fa,
                    density=list(density) if density is not None else None, pose_decoder_kick_in_iter=pose_kick_in,
                    mean_alpha=float(save['alpha'].mean()), max_alpha=float(save['alpha'].max()))
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **save)
        stt = {k: (float(np.abs(v).mean()), float(np.abs(v).max())) for k, v in save.items() if k in
               ('rgb', 'alpha', 'depth', 'offsets')}
        print(name, {k: v for k, v in meta.items() if k != 'frame'}, stt)
        return meta

    os.makedirs(GOLD, exist_ok=True)
    for old in os.listdir(GOLD):
        # (fixtures of the sibling scripts stay: make_golden_patches / _images / _dataset)
        if old.endswith('.npz') and not old.startswith(('patches', 'image_', 'dataset_', 'subject_')):
            os.remove(os.path.join(GOLD, old))
    # ~260 rays of the 512x512 T-pose framing, per-sample tensors kept for the first 64
    FRAME = dict(H=512, W=512, focal_at_512=1250.0, ray_stride=30, pose_seed=0, bgcolor=(0.0, 0.0, 0.0))
    print('rays in golden frame:', scene.synthetic_frame(**FRAME)['rays'].shape)
    metas = {'frame': dict(FRAME, seed=0)}
    metas['eval_s128'] = run_case('eval_s128', 1e7, 128)
    metas['eval_s64'] = run_case('eval_s64', 1e7, 64)
    metas['eval_s256'] = run_case('eval_s256', 1e7, 256)                      # BASELINE config 5 sampling
    metas['tpose_s128'] = run_case('tpose_s128', 1e7, 128, ignore_nr=True)
    metas['iter0_s128'] = run_case('iter0_s128', 0.0, 128)                    # below both kick-ins: cond and PE zeroed
    metas['iter12000_s128'] = run_case('iter12000_s128', 12000.0, 128)        # first Hann band partly open
    metas['iter30000_s128'] = run_case('iter30000_s128', 30000.0, 128)        # bands 0-2 open, band 3 closed
    metas['perturb_s128'] = run_case('perturb_s128', 1e7, 128, perturb=1.0, with_t_rand=True)
    # wild configs: white background (network.py:379 bg blend) and the pose decoder held back (network.py:667)
    metas['whitebg_s128'] = run_case('whitebg_s128', 1e7, 128, frame=dict(bgcolor=(255.0, 255.0, 255.0)))
    metas['posehold_s128'] = run_case('posehold_s128', 12000.0, 128, pose_kick_in=20000)
    # dense medium, camera zoomed on the torso: most rays saturate (transmittance scan over many opaque samples)
    metas['dense_s128'] = run_case('dense_s128', 1e7, 128, frame=dict(focal_at_512=3200.0, ray_stride=31),
                                   density=(45.0, 4.0))
    metas['dense_white_s64'] = run_case('dense_white_s64', 30000.0, 64, perturb=1.0, with_t_rand=True,
                                        frame=dict(focal_at_512=3200.0, ray_stride=31, bgcolor=(255.0, 255.0, 255.0)),
                                        density=(45.0, 4.0))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in base_state.items()}, strict=True)

    # ---- gradients of a scalar loss through the reference (training path, trainer.py:206-220)
    cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = 64, 0.0, False
    net.train()
    for p_ in net.parameters():
        p_.grad = None
    frg = scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, ray_stride=73)
    Rg = frg['rays'].shape[1]
    lw = np.random.RandomState(11).randn(Rg, 5).astype(np.float32)
    out = net(**frame_tensors(frg), iter_val=30000.0)
    loss = (out['rgb'] * torch.from_numpy(lw[:, :3])).sum() + (out['alpha'] * torch.from_numpy(lw[:, 3])).sum() \
        + 0.1 * (out['depth'] * torch.from_numpy(lw[:, 4])).sum()
    loss.backward()
    gsave = {'loss_weights': lw, 'loss': np.float32(loss.item())}
    for k, p_ in net.named_parameters():
        g = p_.grad
        gsave['norm/' + k] = np.float32(0.0 if g is None else g.norm().item())
        if g is not None and g.numel() <= 70000:
            gsave['grad/' + k] = g.numpy().copy()
        elif g is not None:
            gsave['head/' + k] = g.reshape(-1)[:4096].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'grad_s64.npz'), **gsave)
    metas['grad_s64'] = dict(iter_val=30000.0, N_samples=64, ray_stride=73, focal_at_512=1250.0, n_rays=int(Rg),
                             loss=float(loss.item()))
    print('grad golden: rays', Rg, 'loss', loss.item(), 'tensors', sum(1 for k in gsave if k.startswith('norm/')))
    net.eval()

    # ---- reference CPU throughput (indicative)
    cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = 128, 0.0, False
    frb = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0, ray_stride=4)   # 128x128 rays of C2
    tens = frame_tensors(frb)
    times = []
    with torch.no_grad():
        net(**tens, iter_val=1e7)
        for _ in range(3):
            t0 = time.time()
            net(**tens, iter_val=1e7)
            times.append(time.time() - t0)
    nr = frb['rays'].shape[1]
    metas['reference_cpu'] = dict(rays=nr, samples=128, seconds_median=float(np.median(times)),
                                  rays_per_s=float(nr / np.median(times)), threads=torch.get_num_threads(),
                                  torch=torch.__version__, note='reference Network.forward, eval, perturb=0')
    print(metas['reference_cpu'])
    with open(os.path.join(GOLD, 'meta.json'), 'w') as f:
        json.dump(metas, f, indent=1)


if __name__ == '__main__':
    main()
