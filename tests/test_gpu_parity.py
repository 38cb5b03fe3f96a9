"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference-generated golden fixtures.  Needs an MI355X: ``pytest -m gpu``.

fp32 tolerance of the build (both MLP arithmetics), stated once -- SURVEY.md section 8(c):
  per-ray  rgb, alpha            |err| <= 2e-5      (reference's own fp32 noise: 1.0e-5 / 1.5e-5)
           depth                 |err| <= 1e-4 far  (far ~ 5..7; reference noise 9e-5)
  PSNR(rgb) vs reference         >= 90 dB
  per-sample positions           |err| <= 1e-5 where sum w > 1e-3, <= 1e-4 inside the 1e-4 clamp of the
                                 skinning-weight sum (reference noise 6e-5 there)
Per-sample colours sit behind a 2^9 positional-encoding band and are compared
loosely (see tests/test_oracle_golden.py).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_RGB, TOL_ALPHA, TOL_DEPTH_PER_FAR, TOL_XYZ, TOL_XYZ_SOLID = 2e-5, 2e-5, 1e-4, 1e-4, 1e-5


def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2))
    return 200.0 if mse == 0 else -10 * np.log10(mse)


@pytest.fixture(scope='module')
def gpu_net(seeded_params):
    from humannerf_amd.network import Network
    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()}, strict=True)
    return net.to(dev()).eval().deploy_mlps_to_secondary_gpus()


def frame_to_gpu(fr):
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    d = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev()) for k in keys}
    d['head_id'] = torch.tensor(-1)
    return d


# ------------------------------------------------------------------ whole path vs the reference
CASES = ['eval_s128', 'eval_s64', 'eval_s256', 'tpose_s128', 'iter0_s128', 'iter12000_s128', 'iter30000_s128', 'perturb_s128',
         'whitebg_s128', 'posehold_s128', 'dense_s128', 'dense_white_s64']
HEAD = ('cnl_mlp.module.output_linear.0.weight', 'cnl_mlp.module.output_linear.0.bias')


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
@pytest.mark.parametrize('case', CASES)
def test_network_matches_reference_golden(case, mode, gpu_net, golden_case, seeded_params):
    """Network.forward on the GPU vs the outputs of the REFERENCE on identical rays / weights: 12 cases (sampling
    density 64 / 128 / 256, T-pose branch, iterations below / inside / above the Hann window, stratified sampling,
    white background, pose decoder held back, dense medium with saturated rays) x both MLP arithmetics."""
    from humannerf_amd.config import cfg
    m, g, frame, state = golden_case(case)
    cfg.amd.mlp_mode = mode
    cfg.N_samples, cfg.perturb = m['N_samples'], m['perturb']
    cfg.ignore_non_rigid_motions = m['ignore_non_rigid_motions']
    if m['pose_decoder_kick_in_iter'] is not None:
        cfg.pose_decoder.kick_in_iter = m['pose_decoder_kick_in_iter']
    kw = {}
    if 't_rand' in g.files:
        kw['t_rand'] = torch.from_numpy(g['t_rand']).to(dev())
    sd = gpu_net.state_dict()
    try:
        with torch.no_grad():
            for k in HEAD:                                  # the dense cases differ in the sigma row of the head
                sd[k].copy_(torch.from_numpy(state[k]))
            out = gpu_net(**frame_to_gpu(frame), iter_val=m['iter_val'], **kw)
        assert gpu_net.check_f16_range(wait=True) is False      # no golden case comes near the f16 clamp
    finally:
        with torch.no_grad():
            for k in HEAD:
                sd[k].copy_(torch.from_numpy(seeded_params[k]))
        cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = 128, 1.0, False
        cfg.amd.mlp_mode = 'f16x3'
        cfg.pose_decoder.pop('kick_in_iter', None)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    tol_depth = TOL_DEPTH_PER_FAR * frame['far'][:, 0]
    d_depth = np.abs(out['depth'] - g['depth'])
    print(case, mode, 'max err rgb %.2e alpha %.2e depth %.2e (tol %.1e)' % (
        np.abs(out['rgb'] - g['rgb']).max(), np.abs(out['alpha'] - g['alpha']).max(), d_depth.max(), tol_depth.min()),
          '| mean err rgb %.2e; skinning weights max %.2e mean %.2e' % (
        np.abs(out['rgb'] - g['rgb']).mean(), np.abs(out['backward_motion_weights'][:m['keep_rays']] - g['backward_motion_weights']).max(),
        np.abs(out['backward_motion_weights'][:m['keep_rays']] - g['backward_motion_weights']).mean()))
    assert set(out) == {'rgb', 'alpha', 'depth', 'weights_on_rays', 'xyz_on_rays', 'rgb_on_rays', 'cnl_xyz',
                        'cnl_rgb', 'cnl_weight', 'backward_motion_weights', 'offsets'}
    n = m['keep_rays']
    assert out['rgb'].shape == (m['n_rays'], 3) and out['weights_on_rays'].shape == (m['n_rays'], m['N_samples'])
    # dense-medium cases (sigma up to ~400, saturated rays): the reference's own fp32 result is 1.8e-5 (rgb) / 2.4e-5
    # (alpha) away from an fp64 evaluation of the same formulas, so 2e-5 cannot be asked of anybody there: 4e-5
    slack = 2.0 if m['density'] is not None else 1.0
    assert np.abs(out['rgb'] - g['rgb']).max() <= slack * TOL_RGB
    assert np.abs(out['alpha'] - g['alpha']).max() <= slack * TOL_ALPHA
    assert (d_depth <= tol_depth).all()
    assert psnr(out['rgb'], g['rgb']) >= 90.0
    assert np.abs(out['cnl_weight'] - g['cnl_weight']).max() <= slack * TOL_ALPHA
    assert np.abs(out['weights_on_rays'][:n] - g['weights_on_rays']).max() <= slack * TOL_ALPHA
    # positions: tight where the skinning-weight sum is away from its 1e-4 clamp, TOL_XYZ inside the clamp region
    # (ill-conditioned division, see tests/test_oracle_golden.py)
    solid = g['_mask'] > 1e-3
    for k in ('xyz_on_rays', 'offsets'):
        err = np.abs(out[k][:n] - g[k]).max(axis=-1)
        assert err[solid].max() <= TOL_XYZ_SOLID and err.max() <= TOL_XYZ, (k, err[solid].max(), err.max())
    assert np.abs(out['backward_motion_weights'][:n] - g['backward_motion_weights']).max() <= 1e-5
    assert np.abs(out['rgb_on_rays'][:n] - g['rgb_on_rays'])[solid].max() <= 2e-3
    assert np.abs(out['rgb_on_rays'][:n] - g['rgb_on_rays']).max() <= 2e-2
    sel = g['cnl_weight'] > 1e-4
    same = np.abs(out['cnl_xyz'][sel] - g['cnl_xyz'][sel]).max(axis=-1) < 1e-3
    assert same.mean() > 0.97


@pytest.mark.parametrize('B,R,S', [(24, 3001, 128), (24, 17, 7), (7, 513, 64)])
def test_sample_warp_forms_agree_bit_for_bit(B, R, S):
    """K1 exists in four instances (with / without the per-bone weight output, bone count 24 at compile time / read at
    run time).  Round 3: with fp contraction left to the compiler two instances could differ in the last bit of a grid
    coordinate -- 1e-4 of x_skel where the weight sum is small; the kernel now rounds every operation of the reference's
    expressions on its own.  Here: the lean and the diagnostic form give the same z / x_skel / weight sum, bit for bit, on
    samples inside, on the border of and far outside the weight volumes, ragged sample counts included, and the staged
    weight output (LDS transpose, whole-wavefront stores) sums to the weight sum the kernel reports."""
    from humannerf_amd import ops
    g = torch.Generator(device='cpu').manual_seed(5 + B + R)
    G = 32
    rays_o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    near = torch.full((R,), 0.1) + torch.rand(R, generator=g) * 0.1
    far = near + 2.5 + torch.rand(R, generator=g)
    t_rand = torch.rand(R, S, generator=g)
    A = torch.randn(B, 3, 3, generator=g) * 0.3 + torch.eye(3)
    T = torch.randn(B, 3, generator=g) * 0.3
    vol = torch.softmax(torch.randn(B + 1, G, G, G, generator=g) * 2, dim=0).contiguous()
    bmin, bscale = torch.tensor([-0.9, -1.1, -0.7]), torch.tensor([2 / 1.8, 2 / 2.2, 2 / 1.4])
    a = [t.to(dev()).contiguous() for t in (rays_o, rays_d, near, far, t_rand, A, T, vol, bmin, bscale)]
    for tr in (a[4], None):
        z, xs, m, w = ops.sample_warp(a[0], a[1], a[2], a[3], tr, a[5], a[6], a[7], a[8], a[9], S, want_bmw=True)
        z2, xs2, m2, _ = ops.sample_warp(a[0], a[1], a[2], a[3], tr, a[5], a[6], a[7], a[8], a[9], S)
        assert torch.equal(z, z2) and torch.equal(xs, xs2) and torch.equal(m, m2)
        assert 0.2 < float((m > 0).float().mean()) < 0.98           # both regimes present
        s = torch.zeros_like(m)
        for b in range(B):                                          # the kernel's own summation order
            s = s + w[..., b]
        assert torch.equal(s, m)


@pytest.mark.parametrize('which', ['cnl_mlp.module.pts_linears.2.weight', 'non_rigid_mlp.module.block_mlps.2.weight'])
@pytest.mark.parametrize('diag', [True, False])
def test_f16_range_guard_of_inference(which, diag, gpu_net, golden_frame, seeded_params):
    """inference in the default 'f16x3' arithmetic clamps hidden activations at 65504 -- silently,
    until now.  A hidden layer scaled x 3e5 (pre-activations of order 1e5) must raise one frame late (or at the loop's closing
    check) through the status word of the packed image, in both MLPs and on both output paths; with
    cfg.amd.on_f16_range = 'f32' the network switches itself to the exact kernels, whose result is then the fp32 CPU
    oracle's; the seeded weights never trip it (asserted in every golden case above); 'f32' mode has no such limit."""
    from humannerf_amd.config import cfg
    from humannerf_amd.network import ActivationRangeError
    from oracle import oracle
    cfg.perturb, cfg.N_samples, cfg.amd.mlp_mode, cfg.amd.diagnostics = 0., 128, 'f16x3', diag
    data = frame_to_gpu(golden_frame)
    sd = gpu_net.state_dict()
    try:
        with torch.no_grad():
            sd[which].mul_(3.0e5)
            gpu_net(**data, iter_val=1e7)                        # queued: the verdict arrives behind the frame
            with pytest.raises(ActivationRangeError, match='canonical' if which.startswith('cnl') else 'non-rigid'):
                gpu_net.check_f16_range(wait=True)
            assert gpu_net.f16_range_hits >= 1
            # one frame late inside a loop: the second call looks at the first frame's word
            gpu_net(**data, iter_val=1e7)
            torch.cuda.synchronize()
            with pytest.raises(ActivationRangeError):
                gpu_net(**data, iter_val=1e7)
            # fall-back policy: warn, switch to the exact kernels, which then agree with the CPU oracle on these weights
            cfg.amd.on_f16_range = 'f32'
            gpu_net._range_watch = None
            gpu_net(**data, iter_val=1e7)
            with pytest.warns(UserWarning, match="switching this network to 'f32'"):
                assert gpu_net.check_f16_range(wait=True) is True
            assert gpu_net._mlp_mode() == 'f32'
            out = gpu_net(**data, iter_val=1e7)
            assert gpu_net.check_f16_range(wait=True) is False
            state = {k: v.detach().cpu().numpy() for k, v in sd.items()}
            ref = oracle.render(state, golden_frame, iter_val=1e7, N_samples=128)
            # (activations of 1e5: fp32 rounding scales with them and the densities saturate; most rays still agree closely)
            assert float(((out['alpha'].cpu() - ref['alpha']).abs() <= 1e-3).float().mean()) > 0.97
    finally:
        with torch.no_grad():
            sd[which].copy_(torch.from_numpy(seeded_params[which]))
        cfg.amd.on_f16_range, cfg.amd.diagnostics, cfg.perturb = 'raise', True, 1.0
        gpu_net._forced_mode, gpu_net._range_watch, gpu_net._cnl_pack = None, None, None


def test_lean_path_equals_diagnostic_path(gpu_net, golden_frame):
    """hnrf_render_rays_fwd (workspace path, 3 outputs) == the 11-output path."""
    from humannerf_amd import config
    config.cfg.perturb = 0.
    try:
        with torch.no_grad():
            full = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
            config.cfg.amd.diagnostics = False
            lean = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
    finally:
        config.cfg.amd.diagnostics = True
        config.cfg.perturb = 1.0
    assert set(lean) == {'rgb', 'alpha', 'depth'}
    for k in lean:
        assert torch.equal(lean[k], full[k])


# ------------------------------------------------------------------ per-kernel vs the oracle
def test_sample_warp_kernel_edges():
    """K1 vs oracle incl. points outside the volume, near == far, huge coordinates."""
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(1)
    R, S, B, G = 67, 128, 24, 32
    rays_o = rs.uniform(-1, 1, (R, 3)).astype(np.float32)
    rays_d = rs.uniform(-1, 1, (R, 3)).astype(np.float32)
    near = rs.uniform(0.1, 1.0, (R, 1)).astype(np.float32)
    far = near + rs.uniform(0.0, 3.0, (R, 1)).astype(np.float32)
    far[0] = near[0]                      # degenerate interval
    rays_o[1] = [1e6, -1e6, 1e6]          # far outside: every corner zero-padded
    rays_d[2] = 0
    Rs = (np.eye(3)[None] + 0.2 * rs.randn(B, 3, 3)).astype(np.float32)
    Ts = (0.3 * rs.randn(B, 3)).astype(np.float32)
    vol = rs.uniform(0, 1, (B + 1, G, G, G)).astype(np.float32)
    vol[:, :, :4] = 0                     # an exactly-empty slab: sum w == 0 samples
    bmin = np.array([-1.2, -1.4, -0.9], dtype=np.float32)
    bscale = (2.0 / np.array([2.4, 2.8, 1.8])).astype(np.float32)
    t_rand = rs.uniform(0, 1, (R, S)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev())
    for tr in (None, t_rand):
        z, xs, mask, bmw = ops.sample_warp(T(rays_o), T(rays_d), T(near), T(far), None if tr is None else T(tr),
                                           T(Rs), T(Ts), T(vol), T(bmin), T(bscale), S, want_bmw=True)
        zo = oracle.z_values(torch.from_numpy(near), torch.from_numpy(far), S, None if tr is None else torch.from_numpy(tr))
        pts = torch.from_numpy(rays_o)[:, None] + torch.from_numpy(rays_d)[:, None] * zo[:, :, None]
        xo, mo, wo = oracle.sample_motion_fields(pts.reshape(-1, 3), torch.from_numpy(Rs), torch.from_numpy(Ts),
                                                 torch.from_numpy(vol), torch.from_numpy(bmin), torch.from_numpy(bscale))
        assert np.abs(z.cpu().numpy() - zo.numpy()).max() <= 1e-6
        assert np.abs(bmw.cpu().numpy().reshape(-1, B) - wo.numpy()).max() <= 2e-5
        assert np.abs(mask.cpu().numpy().reshape(-1) - mo.numpy()).max() <= 1e-4
        ok = mo.numpy() > 1e-2              # x_skel is ill-conditioned where sum w -> 0
        err = np.abs(xs.cpu().numpy().reshape(-1, 3) - xo.numpy())
        big = np.abs(xo.numpy()) < 1e3
        assert err[ok[:, None] & big].max() <= 2e-4
        assert torch.isfinite(xs).all() and torch.isfinite(mask).all()
        assert (mask.cpu().numpy().reshape(R, S)[1] == 0).all()
        # without the diagnostic output the results are identical
        z2, xs2, mask2, none = ops.sample_warp(T(rays_o), T(rays_d), T(near), T(far), None if tr is None else T(tr),
                                               T(Rs), T(Ts), T(vol), T(bmin), T(bscale), S, want_bmw=False)
        assert none is None and torch.equal(xs, xs2) and torch.equal(mask, mask2)


def _mlp_states(rs):
    from humannerf_amd.seeded import default_shapes
    st = {}
    for k, s in default_shapes().items():
        if k.startswith('cnl_mlp') or k.startswith('non_rigid_mlp'):
            b = np.sqrt(6.0 / sum(s[:2])) * np.sqrt(2) if len(s) == 2 else 0.1
            st[k] = rs.uniform(-b, b, s).astype(np.float32)
    return st


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
@pytest.mark.parametrize('P', [1, 31, 128, 1000, 4133])
def test_canonical_mlp_kernel(P, mode):
    """K3 vs oracle on ragged sample counts (tail masking, one wave, many blocks)."""
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(P)
    st = _mlp_states(rs)
    xyz = rs.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    ws = [st[f'cnl_mlp.module.pts_linears.{i}.weight'] for i in idx] + [st['cnl_mlp.module.output_linear.0.weight']]
    bs = [st[f'cnl_mlp.module.pts_linears.{i}.bias'] for i in idx] + [st['cnl_mlp.module.output_linear.0.bias']]
    T = lambda a: torch.from_numpy(a).to(dev())
    packed = ops.canonical_pack([T(w) for w in ws], [T(b) for b in bs], mode)
    raw = ops.canonical(T(xyz), packed, mode).cpu().numpy()
    ref64 = oracle.canonical_mlp({k: torch.from_numpy(v).double() for k, v in st.items()},
                                 oracle.fourier_pe(torch.from_numpy(xyz).double(), 10)).numpy()
    ref32 = oracle.canonical_mlp({k: torch.from_numpy(v) for k, v in st.items()},
                                 oracle.fourier_pe(torch.from_numpy(xyz), 10)).numpy()
    scale = max(1.0, np.abs(ref64).max())
    e_hip, e_cpu = np.abs(raw - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
    print('canonical', mode, P, 'rel err hip %.2e cpu-fp32 %.2e' % (e_hip, e_cpu))
    assert e_hip <= 2e-5, (e_hip, e_cpu)
    assert e_hip <= 4 * e_cpu + 1e-6, (e_hip, e_cpu)      # as accurate as the CPU fp32 path


def _apply_regime(st, regime, rs):
    """Weight regimes of the non-rigid MLP the kernels must survive in BOTH arithmetics:
      scaled       Xavier weights, last layer x0.1 (offsets of a few cm: a trained network)
      fresh_init   the reference's initialisation: last layer U(+-1e-5), zero bias (mlp_offset.py:60-66) -- every
                   hi = f16(w) of that layer is an f16 SUBNORMAL in the split-f16 kernels
      tiny_hidden  a hidden layer (block_mlps.4) scaled so that its weights and biases are ~1e-6."""
    p = 'non_rigid_mlp.module.block_mlps.'
    if regime == 'scaled':
        st[p + '12.weight'] *= 0.1
    elif regime == 'fresh_init':
        st[p + '12.weight'] = rs.uniform(-1e-5, 1e-5, st[p + '12.weight'].shape).astype(np.float32)
        st[p + '12.bias'] = np.zeros_like(st[p + '12.bias'])
    elif regime == 'tiny_hidden':
        st[p + '4.weight'] *= np.float32(1e-5)
        st[p + '4.bias'] *= np.float32(1e-5)
    else:
        raise ValueError(regime)
    return st


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
@pytest.mark.parametrize('P,iter_val,regime', [(1, 1e7, 'scaled'), (97, 1e7, 'scaled'), (2048, 30000.0, 'scaled'),
                                               (555, 0.0, 'scaled'), (1500, 1e7, 'fresh_init'), (1500, 1.0, 'fresh_init'),
                                               (1500, 1e7, 'tiny_hidden')])
def test_nonrigid_mlp_kernel(P, iter_val, regime, mode):
    """K2 vs an fp64 evaluation.  Error bound relative to the OUTPUT magnitude (the offsets): 1e-5 max|offset|, and the
    accuracy of a torch fp32 evaluation on the CPU as the yardstick."""
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(P + 7)
    st = _apply_regime(_mlp_states(rs), regime, rs)
    x = rs.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    cond = rs.uniform(-0.5, 0.5, (69,)).astype(np.float32)
    if iter_val < 10000:
        cond = cond * 0
    hw = oracle.hann_weights(iter_val, 6, 10000, 50000)
    idx = [0, 2, 4, 6, 8, 10, 12]
    ws = [st[f'non_rigid_mlp.module.block_mlps.{i}.weight'] for i in idx]
    bs = [st[f'non_rigid_mlp.module.block_mlps.{i}.bias'] for i in idx]
    T = lambda a: torch.from_numpy(a).to(dev())
    packed = ops.nonrigid_pack([T(w) for w in ws], [T(b) for b in bs], T(cond), mode)
    xyz, ofs = ops.nonrigid(T(x), hw.to(dev()), packed, mode, want_offsets=True)
    st64 = {k: torch.from_numpy(v).double() for k, v in st.items()}
    xyz64, ofs64 = oracle.non_rigid_mlp(st64, oracle.hann_pe(torch.from_numpy(x).double(), hw.double()),
                                        torch.from_numpy(cond).double()[None], torch.from_numpy(x).double())
    _, ofs32 = oracle.non_rigid_mlp({k: torch.from_numpy(v) for k, v in st.items()},
                                    oracle.hann_pe(torch.from_numpy(x), hw), torch.from_numpy(cond)[None], torch.from_numpy(x))
    top = float(ofs64.abs().max())
    e_hip = float(np.abs(ofs.cpu().numpy() - ofs64.numpy()).max()) / top
    e_cpu = float((ofs32.double() - ofs64).abs().max()) / top
    print('nonrigid', mode, regime, P, 'max|offset| %.2e rel err hip %.2e cpu-fp32 %.2e' % (top, e_hip, e_cpu))
    assert e_hip <= 1e-5, (e_hip, e_cpu)
    assert e_hip <= 8 * e_cpu + 2e-7, (e_hip, e_cpu)          # fp32-class: within a small factor of torch's fp32
    assert np.abs(xyz.cpu().numpy() - xyz64.numpy()).max() <= 2e-7 * 1.3 + 1e-5 * top
    xyz2, none = ops.nonrigid(T(x), hw.to(dev()), packed, mode, want_offsets=False)
    assert none is None and torch.equal(xyz, xyz2)


@pytest.mark.parametrize('S', [2, 64, 100, 128, 256, 300])
def test_composite_kernel(S):
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(S)
    R = 37
    raw = rs.randn(R, S, 4).astype(np.float32) * 3
    raw[..., 3] = rs.randn(R, S) * 30 + 10
    raw[3] = -5.0                                  # an empty ray: all weights zero
    mask = rs.uniform(0, 1.2, (R, S)).astype(np.float32)
    mask[4] = 0
    near = rs.uniform(0.5, 1, (R, 1)).astype(np.float32)
    z = np.sort(near + rs.uniform(0, 3, (R, S)).astype(np.float32), axis=1)
    rays_d = rs.randn(R, 3).astype(np.float32)
    xyz = rs.randn(R, S, 3).astype(np.float32)
    bg = np.array([255., 128., 0.], dtype=np.float32)
    T = lambda a: torch.from_numpy(a).to(dev())
    out = ops.composite(T(raw), T(mask), T(z), T(rays_d), T(xyz), T(bg), diagnostics=True)
    ref = oracle.raw2outputs(*(torch.from_numpy(a).double() for a in (raw, mask, z, rays_d, xyz, bg)))
    for k in ('rgb', 'alpha', 'depth', 'weights_on_rays', 'rgb_on_rays', 'cnl_weight'):
        err = np.abs(out[k].cpu().numpy() - ref[k].numpy()).max()
        assert err <= 3e-6 * max(1.0, float(ref[k].abs().max())), (k, err)
    w = ref['weights_on_rays'].numpy()
    srt = np.sort(w, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-6       # unambiguous argmax
    assert np.array_equal(out['cnl_xyz'].cpu().numpy()[clear], ref['cnl_xyz'].numpy()[clear].astype(np.float32))
    lean = ops.composite(T(raw), T(mask), T(z), T(rays_d), None, T(bg), diagnostics=False)
    for k in lean:
        assert torch.equal(lean[k], out[k])


def test_full_size_properties():
    """BASELINE C2-sized chunk (32768 rays x 128): size-independent properties --
    permutation equivariance over rays, chunking invariance, empty space -> background."""
    from humannerf_amd import ops
    rs = np.random.RandomState(3)
    st = _mlp_states(rs)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    cp = ops.canonical_pack([T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])],
                            [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])])
    idn = [0, 2, 4, 6, 8, 10, 12]
    cond = T(rs.uniform(-0.3, 0.3, 69).astype(np.float32))
    npk = ops.nonrigid_pack([T(st[f'non_rigid_mlp.module.block_mlps.{i}.weight']) for i in idn],
                            [T(st[f'non_rigid_mlp.module.block_mlps.{i}.bias']) for i in idn], cond)
    from humannerf_amd import scene
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0, ray_stride=2)
    R = 32768
    g = {k: T(fr[k]) for k in ('rays', 'near', 'far', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz')}
    o, d, nr, fa = g['rays'][0][:R].contiguous(), g['rays'][1][:R].contiguous(), g['near'][:R], g['far'][:R]
    B = 24
    Rs = T((np.eye(3)[None] + 0.05 * rs.randn(B, 3, 3)).astype(np.float32))
    Ts = T((0.05 * rs.randn(B, 3)).astype(np.float32))
    vol = T(fr['motion_weights_priors'])
    hw = torch.ones(6, device=dev())
    bg = T(np.array([10., 200., 30.], dtype=np.float32))
    args = (Rs, Ts, vol, g['cnl_bbox_min_xyz'], g['cnl_bbox_scale_xyz'], hw, npk, cp, bg, 128)
    a = ops.render_rays(o, d, nr, fa, None, *args)
    assert all(torch.isfinite(v).all() for v in a.values())
    assert float(a['alpha'].min()) >= 0 and float(a['alpha'].max()) <= 1 + 1e-5
    perm = torch.randperm(R, device=dev())
    b = ops.render_rays(o[perm].contiguous(), d[perm].contiguous(), nr[perm].contiguous(), fa[perm].contiguous(), None, *args)
    for k in a:
        assert torch.equal(a[k][perm], b[k])
    h1 = ops.render_rays(o[:10000].contiguous(), d[:10000].contiguous(), nr[:10000].contiguous(), fa[:10000].contiguous(), None, *args)
    for k in a:
        assert torch.equal(a[k][:10000], h1[k])
    empty = ops.render_rays(o, d, nr, fa, None, Rs, Ts, torch.zeros_like(vol), *args[3:])
    assert float(empty['alpha'].abs().max()) == 0
    assert torch.allclose(empty['rgb'], (bg / 255.).expand(R, 3))


def test_sample_culling(gpu_net, golden_frame):
    """Opt-in culling (cfg.amd.cull_eps): eps = 0 is bit-identical to the dense path; eps > 0 stays
    inside the proven bound 2 * S * eps and drops a large share of the samples."""
    from humannerf_amd import config, ops
    cfg = config.cfg
    cfg.perturb, cfg.amd.diagnostics = 0., False
    S = 128
    try:
        for mode in ('f16x3', 'f32'):
            cfg.amd.mlp_mode = mode
            with torch.no_grad():
                cfg.amd.cull_eps = 0.0
                dense = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
                for eps in (1e-9, 1e-7):
                    cfg.amd.cull_eps = eps
                    culled = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
                    for k in ('rgb', 'alpha'):
                        err = float((culled[k] - dense[k]).abs().max())
                        assert err <= 2 * S * eps + 1e-7, (mode, eps, k, err)
                    assert float((culled['depth'] - dense['depth']).abs().max()) <= (2 * S * eps + 1e-7) * 10
    finally:
        cfg.amd.cull_eps, cfg.amd.mlp_mode, cfg.amd.diagnostics, cfg.perturb = 0.0, 'f16x3', True, 1.0
    # the compaction itself: exact index set, count on the device
    rs = np.random.RandomState(0)
    m = rs.uniform(0, 1e-6, 100003).astype(np.float32)
    m[::7] = 0
    idx, count = ops.compact_samples(torch.from_numpy(m).to(dev()), 2e-7)
    n = int(count.item())
    want = np.nonzero(m >= 2e-7)[0]
    assert n == len(want) and np.array_equal(np.sort(idx[:n].cpu().numpy()), want)


def test_render_frames_driver(gpu_net):
    """Frame-sharded driver: image scatter with background fill, 8-bit quantisation, frame order."""
    from humannerf_amd import render, scene
    from humannerf_amd.config import cfg
    frames = [scene.synthetic_frame(H=48, W=48, focal_at_512=1250.0, pose_seed=s, bgcolor=(255., 255., 255.))
              for s in range(3)]
    cfg.amd.diagnostics = False
    try:
        imgs0 = render.render_frames(gpu_net, frames, rank=0, world=2)
        imgs1 = render.render_frames(gpu_net, frames, rank=1, world=2)
    finally:
        cfg.amd.diagnostics = True
    assert sorted(imgs0) == [0, 2] and sorted(imgs1) == [1]
    for i, fr in enumerate(frames):
        img = (imgs0 if i % 2 == 0 else imgs1)[i]
        assert img.shape == (48, 48, 3) and img.dtype == np.uint8
        miss = ~fr['ray_mask'].reshape(48, 48)
        assert (img[miss] == 255).all()                      # background fill where no ray was cast
        assert (img[~miss] < 255).any()
    x = torch.rand(10, 3)
    assert abs(float(render.psnr(x, x + 0.1)) - 20.0) < 1e-4
    # frames that carry the camera instead of host-made rays: generated on the device, same images
    cams = [scene.synthetic_frame(H=48, W=48, focal_at_512=1250.0, pose_seed=s, bgcolor=(255., 255., 255.),
                                  camera_only=True) for s in range(3)]
    cfg.amd.diagnostics = False
    try:
        imgs_c = render.render_frames(gpu_net, cams, rank=0, world=1)
    finally:
        cfg.amd.diagnostics = True
    for i in range(3):
        ref = (imgs0 if i % 2 == 0 else imgs1)[i]
        assert np.abs(imgs_c[i].astype(np.int32) - ref.astype(np.int32)).max() <= 1


def test_ray_chunking_is_invisible(gpu_net, golden_frame):
    """Network._batchify_rays (network.py:330-352): results do not depend on cfg.chunk, also when the
    last chunk is ragged (256 rays in chunks of 50) and in the 11-output diagnostic path."""
    from humannerf_amd.config import cfg
    cfg.perturb = 0.
    try:
        with torch.no_grad():
            cfg.chunk = 32768
            one = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
            cfg.chunk = 50
            many = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
    finally:
        cfg.chunk, cfg.perturb = 32768, 1.0
    assert set(one) == set(many) and len(one) == 11
    for k in one:
        assert torch.equal(one[k], many[k]), k


def test_warp_overlap_on_a_side_stream_changes_nothing(gpu_net, golden_frame):
    """cfg.amd.overlap_warp: K1 of chunk i+1 on a side stream under the MLP kernels of chunk i (hnrf_render_frame_fwd,
    two alternating workspaces, five events) -- same bits as the single-stream sequence, in both output forms, also
    when called back to back (the second frame must not race the first one's side-stream work)."""
    from humannerf_amd.config import cfg
    cfg.perturb, cfg.chunk = 0., 40
    try:
        with torch.no_grad():
            for diag in (True, False):
                cfg.amd.diagnostics = diag
                cfg.amd.overlap_warp = False
                one = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
                cfg.amd.overlap_warp = True
                two = [gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7) for _ in range(3)]
                for t in two:
                    assert set(t) == set(one)
                    for k in one:
                        assert torch.equal(one[k], t[k]), (diag, k)
    finally:
        cfg.chunk, cfg.perturb, cfg.amd.diagnostics, cfg.amd.overlap_warp = 32768, 1.0, True, False


def test_c5_sized_samples_per_ray(gpu_net, golden_frame):
    """BASELINE config 5 uses 256 samples per ray: lean path == diagnostic path at S = 256, and the
    denser sampling converges towards the S = 128 image (same integral)."""
    from humannerf_amd.config import cfg
    cfg.perturb = 0.
    try:
        with torch.no_grad():
            cfg.N_samples = 256
            full = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
            cfg.amd.diagnostics = False
            lean = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
            cfg.N_samples = 128
            s128 = gpu_net(**frame_to_gpu(golden_frame), iter_val=1e7)
    finally:
        cfg.N_samples, cfg.perturb, cfg.amd.diagnostics = 128, 1.0, True
    assert full['weights_on_rays'].shape[-1] == 256
    for k in lean:
        assert torch.equal(lean[k], full[k])
    assert float((full['alpha'] - s128['alpha']).abs().mean()) < 0.02


@pytest.mark.parametrize('H,W,focal', [(64, 64, 1700.0), (120, 96, 1250.0), (512, 512, 1250.0), (37, 53, 600.0)])
def test_ray_generation_kernel(H, W, focal):
    """hnrf_gen_rays vs the numpy restatement of get_rays_from_KRT + rays_intersect_3d_bbox (scene.py, itself
    pinned to the reference's helpers by tests/golden): same hit mask, rays in pixel order, near/far.
    Tolerance: the float32 ray arithmetic may differ from numpy's BLAS by an ulp (summation order), near/far are
    float64 functions of those rays."""
    from humannerf_amd import ops, scene
    J = scene.TPOSE_JOINTS
    mn, mx = (J.min(0) - 0.3).astype(np.float32), (J.max(0) + 0.3).astype(np.float32)
    K, E = scene.tpose_camera(np.array([W, H], dtype=np.float32), 6.0, focal * H / 512.0)
    ro, rd = scene.get_rays_from_KRT(H, W, K, E[:3, :3], E[:3, 3])
    ro, rd = ro.reshape(-1, 3).astype(np.float32), rd.reshape(-1, 3).astype(np.float32).copy()
    near, far, hit = scene.rays_intersect_3d_bbox(np.stack([mn, mx]), ro, rd)
    got = ops.gen_rays(K, E, mn, mx, H, W)
    gm = got['ray_mask'].cpu().numpy()
    # rays grazing an edge may flip with an ulp of difference in d: none do on these cameras
    assert gm.shape == hit.shape and (gm != hit).sum() == 0
    assert got['rays'].shape == (3, int(hit.sum()), 3)
    np.testing.assert_allclose(got['rays'][0].cpu().numpy(), ro[hit], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got['rays'][1].cpu().numpy(), rd[hit], rtol=2e-6, atol=2e-6)
    assert torch.equal(got['rays'][1], got['rays'][2])
    np.testing.assert_allclose(got['near'][:, 0].cpu().numpy(), near.astype(np.float32), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(got['far'][:, 0].cpu().numpy(), far.astype(np.float32), rtol=2e-6, atol=2e-6)
    if focal < 1700.0:
        assert 0 < hit.sum() < H * W                          # the compaction really dropped pixels


def test_c5_full_frame_is_chunk_and_neighbour_independent(seeded_params):
    """BASELINE config 5 at full size (1024 x 1024 rays x 256 samples = 268 M samples, rays generated on the
    device): every 4099-th ray rendered alone gives bit-identical rgb / alpha / depth -- a ray's result does not
    depend on which chunk it travels in or on its neighbours (size-independent property of the path)."""
    from humannerf_amd import ops, scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    d = dev()
    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(d).eval()
    fr = scene.synthetic_frame(H=1024, W=1024, focal_at_512=1700.0, camera_only=True)
    rays = ops.gen_rays(fr['K'], fr['E'], fr['cnl_bbox_min_xyz'], fr['cnl_bbox_max_xyz'], 1024, 1024)
    assert rays['rays'].shape[1] == 1024 * 1024                       # this camera sees the box in every pixel
    keys = ['dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz',
            'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(d) for k in keys}
    data.update(rays=rays['rays'], near=rays['near'], far=rays['far'])
    cfg.perturb, cfg.N_samples, cfg.amd.diagnostics = 0., 256, False
    try:
        with torch.no_grad():
            out = net(**data, iter_val=1e7)
            sel = torch.arange(0, 1024 * 1024, 4099, device=d)
            sub = dict(data, rays=data['rays'][:, sel].contiguous(), near=data['near'][sel].contiguous(),
                       far=data['far'][sel].contiguous())
            alone = net(**sub, iter_val=1e7)
    finally:
        cfg.perturb, cfg.N_samples, cfg.amd.diagnostics = 1.0, 128, True
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.isfinite(out[k]).all()
        assert torch.equal(out[k][sel], alone[k]), k
    assert 0.05 < float(out['alpha'].mean()) < 0.999


@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
def test_early_ray_termination_is_bounded(gpu_net, mode):
    """Opt-in early ray termination (cfg.amd.term_eps, lean path): front-to-back slabs of 32 samples, rays whose
    transmittance fell below term_eps are not evaluated further.  The result moves by at most ~term_eps, the number
    of evaluated samples drops, and a vanishing threshold reproduces the dense result up to the slab-wise product
    order."""
    from humannerf_amd import ops, scene
    from humannerf_amd.config import cfg
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0, ray_stride=4)       # 16 384 rays
    data = frame_to_gpu(fr)
    cfg.perturb, cfg.amd.diagnostics, cfg.amd.mlp_mode = 0., False, mode
    try:
        with torch.no_grad():
            dense = gpu_net(**data, iter_val=1e7)
            res = {}
            for eps in (1e-3, 1e-30):
                cfg.amd.term_eps = eps
                res[eps] = gpu_net(**data, iter_val=1e7)
            cfg.amd.term_eps, cfg.amd.cull_eps = 1e-3, 1e-9
            both = gpu_net(**data, iter_val=1e7)
    finally:
        cfg.perturb, cfg.amd.diagnostics, cfg.amd.mlp_mode, cfg.amd.term_eps, cfg.amd.cull_eps = 1.0, True, 'f16x3', 0.0, 0.0
    for k, tol in (('rgb', 1.5e-3), ('alpha', 1.5e-3)):
        assert float((res[1e-3][k] - dense[k]).abs().max()) <= tol, k
        assert float((both[k] - dense[k]).abs().max()) <= tol, k
        assert float((res[1e-30][k] - dense[k]).abs().max()) <= 2e-6, k
    assert float((res[1e-30]['depth'] - dense['depth']).abs().max()) <= 2e-5
    assert float((res[1e-3]['rgb'] - dense['rgb']).abs().max()) > 0.0          # the cut really happened


def test_early_ray_termination_evaluates_fewer_samples():
    """hnrf_render_rays_term_fwd reports the number of samples that went through the MLPs."""
    from humannerf_amd import ops
    d = dev()
    rs = np.random.RandomState(2)
    R, S, B, G = 512, 128, 24, 32
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(d)
    st = _mlp_states(rs)
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    cw = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
    cb = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
    cb[-1] = cb[-1].clone(); cb[-1][3] += 40.0                                   # dense medium: rays saturate quickly
    names = [f'non_rigid_mlp.module.block_mlps.{i}' for i in (0, 2, 4, 6, 8, 10, 12)]
    nw, nb = [T(st[n + '.weight']) for n in names], [T(st[n + '.bias']) for n in names]
    rays_o = T(rs.uniform(-0.2, 0.2, (R, 3)) + np.array([0, 0, -3.0])); rays_d = T(np.concatenate([rs.uniform(-0.25, 0.25, (R, 2)), np.ones((R, 1))], 1))
    near, far = T(rs.uniform(2.0, 2.3, R)), T(rs.uniform(3.6, 4.0, R))
    Rs = T(np.tile(np.eye(3), (B, 1, 1))); Ts = T(rs.uniform(-0.1, 0.1, (B, 3)))
    vol = T(rs.uniform(0.0, 0.08, (B + 1, G, G, G)))
    bmin, bscale = T(np.full(3, -1.2)), T(np.full(3, 2.0 / 2.4))
    hann, bg = torch.ones(6, device=d), T(np.array([255., 128., 0.]))
    nrp = ops.nonrigid_pack(nw, nb, T(np.zeros(69)), 'f16x3'); cnp = ops.canonical_pack(cw, cb, 'f16x3')
    dense = ops.render_rays(rays_o, rays_d, near, far, None, Rs, Ts, vol, bmin, bscale, hann, nrp, cnp, bg, S, 'f16x3')
    out = ops.render_rays_term(rays_o, rays_d, near, far, None, Rs, Ts, vol, bmin, bscale, hann, nrp, cnp, bg, S,
                               'f16x3', term_eps=1e-4, want_count=True)
    n_eval = int(out['evaluated'].item())
    assert 0 < n_eval < R * S // 2, n_eval
    assert float((out['rgb'] - dense['rgb']).abs().max()) <= 2e-4
    assert float((out['alpha'] - dense['alpha']).abs().max()) <= 2e-4


def test_image_unpack_on_device_matches_reference_bytes(golden_dir):
    """render.unpack_to_image on the GPU vs what the reference's run.unpack_to_image returned (tests/golden/
    image_unpack.npz): rgb, alpha and truth images byte for byte, three background colours."""
    from tests.test_image_side import check_unpack
    check_unpack(np.load(os.path.join(golden_dir, 'image_unpack.npz')), dev())


def test_render_frames_writes_pngs_and_truth(gpu_net, tmp_path):
    """run.py:68-157 end to end on synthetic frames: render -> device-side unpack -> pinned async D2H -> threaded PNG
    writer; the images on disk are the images returned, the truth image is the scattered ``target_rgbs``, and what was
    rendered equals Network.forward's rgb quantised like the reference does."""
    from PIL import Image
    from humannerf_amd import render, scene
    from humannerf_amd.config import cfg
    frames = [scene.synthetic_frame(H=64, W=48, focal_at_512=1250.0, pose_seed=s, bgcolor=(30., 120., 250.)) for s in range(4)]
    rs = np.random.RandomState(2)
    for fr in frames:
        fr['target_rgbs'] = rs.rand(fr['rays'].shape[1], 3).astype(np.float32)
    writer = render.ImageWriter(str(tmp_path), 'freeview', workers=2)
    got = {}

    def on_image(i, rgb8, a8, t8):
        got[i] = (rgb8, a8, t8)
        writer.append(np.concatenate([rgb8, t8, a8], axis=1), img_name='%06d' % i)      # run.py:131-140 layout
    cfg.amd.diagnostics = False
    try:
        imgs = render.render_frames(gpu_net, frames, on_image=on_image, show_truth=True)
        with torch.no_grad():
            cfg.perturb = 0.
            direct = gpu_net(**frame_to_gpu(frames[2]), iter_val=float(cfg.eval_iter))
    finally:
        cfg.amd.diagnostics, cfg.perturb = True, 1.0
    writer.finalize()
    assert sorted(imgs) == [0, 1, 2, 3] and sorted(got) == [0, 1, 2, 3]
    for i, fr in enumerate(frames):
        H, W = 64, 48
        on_disk = np.asarray(Image.open(tmp_path / 'freeview' / ('%06d.png' % i)))
        assert on_disk.shape == (H, 3 * W, 3)
        assert np.array_equal(on_disk[:, :W], imgs[i]) and np.array_equal(on_disk[:, W:2 * W], got[i][2])
        truth = np.full((H * W, 3), np.array([30., 120., 250.], np.float32) / 255., np.float32)
        truth[fr['ray_mask']] = fr['target_rgbs']
        assert np.array_equal(got[i][2], render.to_8b_image(truth.reshape(H, W, 3)))
    ref = np.full((64 * 48, 3), np.array([30., 120., 250.], np.float32) / 255., np.float32)
    ref[frames[2]['ray_mask']] = direct['rgb'].cpu().numpy()
    assert np.array_equal(imgs[2], render.to_8b_image(ref.reshape(64, 48, 3)))


def test_subject_directory_renders_end_to_end(gpu_net, golden_dir):
    """A prepared subject directory (the reference's on-disk layout) -> dataset.Subject -> render_frames: the
    camera-only route (rays generated on the device against the POSED skeleton's bbox) gives the image of the
    reference's numpy route (rays / near / far made on the host exactly like its Dataset does), to 1 LSB."""
    from humannerf_amd import dataset, render
    from humannerf_amd.config import cfg
    subj = dataset.Subject(os.path.join(golden_dir, 'subject_synth'))
    cams = [subj.movement_frame(i, bgcolor=(255., 255., 255.), image_size=(64, 48)) for i in range(3)]
    host = [subj.movement_frame(i, bgcolor=(255., 255., 255.), host_rays=True, image_size=(64, 48)) for i in range(3)]
    cfg.amd.diagnostics, cfg.N_samples = False, 64
    try:
        a = render.render_frames(gpu_net, cams)
        b = render.render_frames(gpu_net, host)
    finally:
        cfg.amd.diagnostics, cfg.N_samples = True, 128
    for i in range(3):
        assert a[i].shape == (64, 48, 3)
        assert np.abs(a[i].astype(np.int32) - b[i].astype(np.int32)).max() <= 1
        assert (a[i] < 250).any()                            # the body is in the picture


def test_run_modes_on_a_subject_directory(gpu_net, golden_dir, tmp_path):
    """humannerf_amd.run: run.py's movement / freeview / tpose loops (run.py:67-183, 212-445) over the synthetic subject
    directory -- files in the reference's layout, truth panel next to the render, PSNR text files, and the movement
    render equal to the frame rendered directly from host-side rays."""
    from PIL import Image
    from humannerf_amd import dataset, render, run
    from humannerf_amd.config import cfg
    subj = dataset.Subject(os.path.join(golden_dir, 'subject_synth'))
    old = (cfg.amd.diagnostics, cfg.N_samples, cfg.get('show_truth', False), cfg.get('show_alpha', False), cfg.bgcolor)
    cfg.amd.diagnostics, cfg.N_samples, cfg.show_alpha, cfg.bgcolor = False, 64, False, [255., 255., 255.]
    try:
        mv = run.run_movement(gpu_net, subj, logdir=str(tmp_path), metrics=['psnr'])
        fv = run.run_freeview(gpu_net, subj, frame_idx=1, total_frames=4, logdir=str(tmp_path))
        cfg.show_alpha = True
        tp = run.run_tpose(gpu_net, subj, total_frames=3, image_size=64, logdir=str(tmp_path))
        assert cfg.ignore_non_rigid_motions is False                   # restored
        host = [subj.movement_frame(i, host_rays=True, load_image=True) for i in range(3)]
        direct = render.render_frames(gpu_net, host, show_truth=True)
    finally:
        cfg.amd.diagnostics, cfg.N_samples, cfg.show_truth, cfg.show_alpha, cfg.bgcolor = old
    root = tmp_path / 'latest'
    names = sorted(os.listdir(root / 'movement'))
    assert names == ['frame_000003.png', 'frame_000010.png', 'frame_000042.png']
    for i, n in enumerate(names):
        im = np.asarray(Image.open(root / 'movement' / n))
        assert im.shape == (64, 2 * 48, 3)                             # render | truth
        assert np.abs(im[:, :48].astype(np.int32) - direct[i].astype(np.int32)).max() <= 1
        assert np.array_equal(im[:, :48], mv['images'][i])
    per_img = open(root / 'movement-metrics.perimg.txt').read()
    assert 'frame_000010: psnr-' in per_img and mv['metrics']['psnr'] > 3.0
    assert open(root / 'movement-metrics.average.txt').read().splitlines()[-1].startswith('p:')
    assert sorted(os.listdir(root / 'freeview_1')) == ['%06d.png' % i for i in range(4)]
    assert np.asarray(Image.open(root / 'freeview_1' / '000002.png')).shape == (64, 48, 3)
    assert not np.array_equal(fv['images'][0], fv['images'][2])        # the camera moved
    t = np.asarray(Image.open(root / 'tpose' / '000001.png'))
    assert t.shape == (64, 2 * 64, 3)                                  # render | alpha
    assert (t[:, :64] < 250).any() and t[:, 64:].max() > 100 and len(tp['images']) == 3


def test_device_frame_cache_equals_host_route(golden_dir):
    """dataset.DeviceFrameCache.train_batch (frame constants resident in HBM, rays from hnrf_gen_rays, crops and
    compositing on the device) returns what to_device(Subject.train_frame) returns for the same state of the global numpy
    generator: same patches, masks and targets, rays / near / far to the ray generator's 2e-6."""
    from humannerf_amd import dataset
    from humannerf_amd.config import cfg
    dev = torch.device('cuda:0')
    subj = dataset.Subject(os.path.join(golden_dir, 'subject_synth'))
    cache = dataset.DeviceFrameCache(subj, dev)
    old = (cfg.patch.N_patches, cfg.patch.size)
    cfg.patch.N_patches, cfg.patch.size = 4, 16
    try:
        for idx, seed in ((0, 1), (2, 2), (2, 3), (1, 4)):
            np.random.seed(seed)
            want = subj.train_frame(idx)
            np.random.seed(seed)
            got = cache.train_batch(idx)
            assert got['frame_name'] == want['frame_name'] and got['img_width'] == want['img_width']
            assert np.array_equal(got['patch_div_indices'].numpy(), want['patch_div_indices'])
            assert np.array_equal(got['patch_masks'].cpu().numpy(), want['patch_masks'])
            assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask'])
            assert np.array_equal(got['bgcolor'].cpu().numpy(), want['bgcolor'])
            for k in ('target_patches', 'target_rgbs'):
                assert got[k].dtype == torch.float32
                assert np.abs(got[k].cpu().numpy() - want[k]).max() <= 6e-8, k
            for k in ('rays', 'near', 'far'):
                assert got[k].shape == want[k].shape
                np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=2e-6, atol=2e-6, err_msg=k)
            for k in ('dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz',
                      'cnl_bbox_scale_xyz'):
                assert np.array_equal(got[k].cpu().numpy(), want[k]), k
        assert len(cache.entries) == 3 and cache.bytes > 0
    finally:
        cfg.patch.N_patches, cfg.patch.size = old


def test_frames_without_rays_render_as_background(gpu_net):
    """A camera that does not see the subject's bbox yields zero rays.  The reference's chunk loop (network.py:330-352)
    has nothing to concatenate then and raises; here Network.forward returns its keys empty and the render loop delivers
    the background image -- on both output paths, between ordinary frames."""
    from humannerf_amd import render, scene
    from humannerf_amd.config import cfg
    cams = [scene.synthetic_frame(H=64, W=64, focal_at_512=1250.0, pose_seed=i, camera_only=True, bgcolor=(30., 60., 90.))
            for i in range(3)]
    E = cams[1]['E'].copy()
    E[:3, 3] += np.array([50., 0., 0.], dtype=np.float32)
    cams[1] = dict(cams[1], E=E)
    old = (cfg.N_samples, cfg.perturb, cfg.amd.diagnostics)
    cfg.N_samples, cfg.perturb = 64, 0.
    try:
        for diag in (True, False):
            cfg.amd.diagnostics = diag
            imgs = render.render_frames(gpu_net, cams, device=dev())
            assert sorted(imgs) == [0, 1, 2] and imgs[1].shape == (64, 64, 3)
            assert (imgs[1] == np.array([30, 60, 90], dtype=np.uint8)).all()            # background only
            assert (imgs[0] != imgs[1]).any() and (imgs[2] != imgs[1]).any()
        fr = scene.synthetic_frame(H=64, W=64, focal_at_512=1250.0)
        data = frame_to_gpu(fr)
        data['rays'], data['near'], data['far'] = data['rays'][:, :0].contiguous(), data['near'][:0].contiguous(), data['far'][:0].contiguous()
        cfg.amd.diagnostics = True
        with torch.no_grad():
            out = gpu_net(**data, iter_val=1e7)
        assert len(out) == 11 and out['rgb'].shape == (0, 3) and out['weights_on_rays'].shape == (0, 64)
        assert out['backward_motion_weights'].shape == (0, 64, 24)
    finally:
        cfg.N_samples, cfg.perturb, cfg.amd.diagnostics = old
