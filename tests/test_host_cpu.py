"""CPU-side checks (no GPU): checkpoint-key contract, C-ABI exports, config
semantics, scene helpers, loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract():
    """Key names and shapes of the reference checkpoint."""
    from humannerf_amd.network import Network
    from humannerf_amd.seeded import default_shapes
    net = Network()
    sd = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert sd == default_shapes()
    assert sum(v.numel() for v in net.state_dict().values()) == 64417381
    # ... of which the optimizer sees 35 057 253: the decoder's first transposed convolution (1x1x1 input) trains the central
    # 2x2x2 of its 4x4x4 taps only -- the others never receive a gradient in the reference either (network._PointDeconv)
    k0 = 'mweight_vol_decoder.decoder.block_conv.0.weight'
    assert dict(net.named_parameters())[k0].shape == (1024, 512, 2, 2, 2)
    assert sum(p.numel() for p in net.parameters()) == 64417381 - 1024 * 512 * 56
    # optimizer routes learning rates by these substrings (optimizer.py:9-34)
    names = [n for n, _ in net.named_parameters()]
    for sub in ('mweight_vol_decoder', 'pose_decoder', 'non_rigid_mlp', 'cnl_mlp'):
        assert any(sub in n for n in names)
    assert net.deploy_mlps_to_secondary_gpus() is net


def test_init_distribution():
    from humannerf_amd.network import Network
    net = Network()
    sd = net.state_dict()
    assert sd['non_rigid_mlp.module.block_mlps.12.weight'].abs().max() <= 1e-5
    assert sd['pose_decoder.block_mlps.8.weight'].abs().max() <= 1e-5
    w = sd['cnl_mlp.module.pts_linears.2.weight']
    bound = np.sqrt(2.0) * np.sqrt(2.0 / 512) * np.sqrt(3.0)
    assert w.abs().max() <= bound and w.abs().max() > 0.95 * bound
    assert sd['cnl_mlp.module.pts_linears.2.bias'].abs().max() == 0
    k = sd['mweight_vol_decoder.decoder.block_conv.0.weight']
    assert torch.equal(k[:, :, 0::2, 0::2, 0::2], k[:, :, 1::2, 1::2, 1::2])


def test_abi_exports_every_declared_symbol():
    """libhnrf.so loads and exports exactly what include/hnrf.h declares."""
    from humannerf_amd import _lib
    with open(os.path.join(ROOT, 'include', 'hnrf.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    declared = set(re.findall(r'\b(hnrf_[a-z0-9_]+)\s*\(', text))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.hnrf_abi_version() == 13
    assert lib.hnrf_canonical_packed_bytes(0) > 1 << 20
    assert lib.hnrf_render_workspace_bytes(4, 128) >= 4 * 128 * 44


def test_abi_argument_errors_do_not_need_a_gpu():
    from humannerf_amd import _lib
    lib = _lib.load()
    rc = lib.hnrf_canonical_fwd(None, None, 0, 10, None, None)
    assert rc == -1 and b'null pointer' in lib.hnrf_last_error()
    rc = lib.hnrf_canonical_fwd(1, 16, 7, 10, 16, None)
    assert rc == -2
    # every entry point validates before it launches: the round-2 additions
    import ctypes
    err = lambda: lib.hnrf_last_error().decode()
    assert lib.hnrf_motion_basis_fwd(None, None, None, 24, None, None, None, None) == -1 and 'null pointer' in err()
    assert lib.hnrf_motion_basis_fwd(8, 8, 8, 23, 8, 8, None, None) == -2 and '23 bones' in err()
    assert lib.hnrf_refined_motion_basis_bwd(8, 8, None, 8, 8, 8, 24, 8, 8, 8, 8, None) == -1
    dims = (ctypes.c_int * 3)(69, 300, 69)
    ptrs = (ctypes.c_void_p * 2)(8, 8)
    assert lib.hnrf_pose_mlp_fwd(8, ptrs, ptrs, dims, 2, 8, 8, None) == -2 and 'width 300' in err()
    assert lib.hnrf_pose_mlp_fwd(8, ptrs, ptrs, dims, 12, 8, 8, None) == -2 and '12 layers' in err()
    assert lib.hnrf_pose_mlp_saved_bytes(5) == (4 + 16) * 256 * 4 and lib.hnrf_pose_mlp_saved_bytes(0) == 0
    assert lib.hnrf_motion_basis_saved_bytes() == 2 * 24 * 16 * 8
    assert lib.hnrf_mlp_dw_h(None, 0, None, 0, 10, 256, 256, 0, None, None, 0, None, None, 0, None) == -1
    assert lib.hnrf_mlp_dw_h(8, 256, 8, 256, 10, 200, 256, 0, None, 8, 256, None, 8, 1 << 30, None) == -2 and 'not built' in err()
    assert lib.hnrf_mlp_dw_h(8, 256, 8, 256, 10, 256, 256, 4, None, 8, 256, None, 8, 1 << 30, None) == -1 and 'bad layout' in err()
    assert lib.hnrf_mlp_dw_h_workspace_bytes(786432, 256, 256) > 0 and lib.hnrf_mlp_dw_h_workspace_bytes(10, 200, 256) == 0


def test_hot_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from humannerf_amd import ops, _lib
    with pytest.raises(_lib.HnrfError):
        ops.canonical(torch.zeros(8, 3), torch.zeros(16))


def test_cfg_semantics():
    from humannerf_amd.config import get_cfg_defaults
    c = get_cfg_defaults()
    assert c.N_samples == 128 and c.chunk == 32768 and c.netchunk_per_gpu == 300000
    c.merge_from_list(['N_samples', '64', 'non_rigid_motion_mlp.kick_in_iter', '100'])
    assert c.N_samples == 64 and c.non_rigid_motion_mlp.kick_in_iter == 100
    with pytest.raises(KeyError):
        c.merge_from_list(['nope', '1'])
    c.perturb = 0.
    assert c.clone().perturb == 0.


def test_scene_frame_shapes():
    from humannerf_amd import scene
    fr = scene.synthetic_frame(H=64, W=64, focal_at_512=1700.0)
    assert fr['rays'].shape == (3, 64 * 64, 3)        # every pixel hits the bbox
    assert fr['near'].shape == (4096, 1) and (fr['far'] > fr['near']).all()
    p = fr['motion_weights_priors']
    assert p.shape == (25, 32, 32, 32) and abs(p.sum(0) - 1).max() < 1e-5
    assert fr['dst_Rs'].shape == (24, 3, 3) and fr['cnl_gtfms'].shape == (24, 4, 4)


def test_hann_weights_match_oracle():
    from humannerf_amd.network import hann_window_weights
    from oracle import oracle
    for it in (0, 9999, 10000, 12345.0, 30000, 50000, 1e7):
        a = hann_window_weights(it, 6, 10000, 50000)
        b = oracle.hann_weights(it, 6, 10000, 50000)
        assert torch.equal(a, b)


def test_optimizer_groups_and_lr_schedule():
    """optimizer.py:12-43 and exp_decay.py:7-16 semantics."""
    from humannerf_amd.network import Network
    from humannerf_amd import train
    net = Network()
    opt = train.build_optimizer(net)
    by_name = {}
    for g in opt.param_groups:
        by_name.setdefault(g['name'], []).append(g['lr'])
    assert set(by_name['mweight_vol_decoder']) == {5e-5} and set(by_name['non_rigid_mlp']) == {5e-5}
    assert set(by_name['pose_decoder']) == {5e-5}
    assert by_name['cnl_mlp.module.pts_linears.0.weight'] == [5e-4]
    assert sum(len(g['params']) for g in opt.param_groups) == 55
    train.update_lr(opt, 500000)
    for g in opt.param_groups:
        base = 5e-5 if g['name'] in ('mweight_vol_decoder', 'non_rigid_mlp', 'pose_decoder') else 5e-4
        assert abs(g['lr'] - base * 0.1) < 1e-12


def test_patch_unpack_and_loss():
    from humannerf_amd import train
    rs = np.random.RandomState(0)
    masks = torch.from_numpy(rs.rand(2, 4, 4) > 0.4)
    n0, n1 = int(masks[0].sum()), int(masks[1].sum())
    rgbs = torch.from_numpy(rs.rand(n0 + n1, 3).astype(np.float32))
    targets = torch.from_numpy(rs.rand(2, 4, 4, 3).astype(np.float32))
    bg = torch.tensor([0.2, 0.4, 0.6])
    img = train.unpack_patches(rgbs, masks, bg, targets, [0, n0, n0 + n1])
    assert torch.equal(img[0][masks[0]], rgbs[:n0]) and torch.equal(img[1][masks[1]], rgbs[n0:])
    assert torch.allclose(img[0][~masks[0]], bg.expand(int((~masks[0]).sum()), 3))
    loss, parts = train.image_loss(img, targets, None)
    assert set(parts) == {'mse'} and abs(float(loss) - 0.2 * float(((img - targets) ** 2).mean())) < 1e-7


def test_conv_transpose_as_batched_gemm_matches_torch():
    """network.conv_transpose3d_k4s2p1 (the weight-volume decoder's layers, network_util.py:12-50) against
    F.conv_transpose3d in fp64: output and all three gradients."""
    import torch.nn.functional as F
    from humannerf_amd.network import conv_transpose3d_k4s2p1
    torch.manual_seed(0)
    for cin, cout, dims in ((6, 5, (1, 1, 1)), (4, 3, (2, 3, 2)), (8, 25, (4, 4, 4))):
        x = torch.randn(1, cin, *dims, dtype=torch.float64, requires_grad=True)
        w = torch.randn(cin, cout, 4, 4, 4, dtype=torch.float64, requires_grad=True)
        b = torch.randn(cout, dtype=torch.float64, requires_grad=True)
        ref = F.conv_transpose3d(x, w, b, stride=2, padding=1)
        got = conv_transpose3d_k4s2p1(x, w, b)
        assert got.shape == ref.shape and (got - ref).abs().max() < 1e-12
        g = torch.randn_like(ref)
        for a, r in zip(torch.autograd.grad(got, (x, w, b), g), torch.autograd.grad(ref, (x, w, b), g)):
            assert (a - r).abs().max() < 1e-11


def test_patch_sampler_matches_reference(golden_dir):
    """scene.sample_patch_rays against what the reference's Dataset.get_patch_ray_indices returned for the same
    masks and the same seed of the global numpy generator (tests/golden/patches_s96.npz, oracle/make_golden_patches.py):
    identical ray indices, patch masks (with holes at the rim of the bbox), window corners and split points."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    from humannerf_amd import scene
    g = np.load(os.path.join(golden_dir, 'patches_s96.npz'))
    H = W = 96
    fr = scene.synthetic_frame(H=H, W=W, focal_at_512=1250.0)
    yy, xx = np.mgrid[0:H, 0:W]
    rim = (yy - H * 0.5) ** 2 / (H * 0.46) ** 2 + (xx - W * 0.5) ** 2 / (W * 0.30) ** 2 < 1.0
    ray_mask = fr['ray_mask'].astype(bool) & rim.reshape(-1)
    subject = ((yy - H * 0.5) ** 2 / (H * 0.36) ** 2 + (xx - W * 0.5) ** 2 / (W * 0.16) ** 2 < 1.0) & ray_mask.reshape(H, W)
    np.random.seed(20240)
    sel, info, div = scene.sample_patch_rays(ray_mask, subject, ray_mask.reshape(H, W).copy(), 6, 20, H, W,
                                             subject_ratio=float(g['subject_ratio']))
    assert np.array_equal(sel, g['select_inds']) and np.array_equal(div, g['div'])
    assert np.array_equal(info['mask'], g['mask'])
    assert np.array_equal(info['xy_min'], g['xy_min']) and np.array_equal(info['xy_max'], g['xy_max'])
    assert (~g['mask']).any() and div[-1] < 6 * 400          # the fixture really has partly covered windows


def test_grouped_adam_is_torch_adam_with_fewer_launches():
    """train.GroupedAdam: same numbers as torch.optim.Adam over the reference's one-group-per-tensor layout, None
    gradients skipped (their step counters do not advance), state_dict in torch's standard form."""
    from humannerf_amd.train import GroupedAdam
    torch.manual_seed(0)
    ps = [torch.randn(5, 3), torch.randn(7), torch.randn(2, 2)]
    a = [p.clone().requires_grad_() for p in ps]
    b = [p.clone().requires_grad_() for p in ps]
    ga = GroupedAdam([{'params': [a[0]], 'lr': 1e-2, 'name': 'x'}, {'params': [a[1]], 'name': 'y'},
                      {'params': [a[2]], 'lr': 1e-2, 'name': 'z'}], lr=1e-3)
    gb = torch.optim.Adam([{'params': [b[0]], 'lr': 1e-2}, {'params': [b[1]]}, {'params': [b[2]], 'lr': 1e-2}], lr=1e-3)
    for it in range(6):
        for x, y in zip(a, b):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        if it == 2:
            a[1].grad = b[1].grad = None
        ga.step()
        gb.step()
    for x, y in zip(a, b):
        assert (x - y).abs().max() <= 1e-7
    # every stepped parameter's version counter moved with every step: the network's packed-weight / weight-volume caches
    # are keyed by it (torch._fused_adam_ alone leaves it where it was: renders after training used stale weights)
    assert a[0]._version >= 6 and a[1]._version >= 5 and a[2]._version >= 6
    sd = ga.state_dict()
    assert float(sd['state'][0]['step']) == 6.0 and float(sd['state'][1]['step']) == 5.0
    assert [g['name'] for g in sd['param_groups']] == ['x', 'y', 'z']
    fresh = GroupedAdam([{'params': [p.clone().requires_grad_()], 'name': n} for p, n in zip(ps, 'xyz')], lr=1e-3)
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.state_dict()['state'][2]['exp_avg'], sd['state'][2]['exp_avg'])


def test_unbuilt_config_branches_raise():
    """every switch of the reference constructors that selects a branch outside the hot path must
    raise instead of being ignored (a checkpoint trained with it would otherwise load non-strictly and render wrong)."""
    from humannerf_amd.config import get_cfg_defaults
    from humannerf_amd.network import check_config_is_built
    check_config_is_built(get_cfg_defaults())
    for path, val in [('canonical_mlp.mlp_depth_plus', 2), ('non_rigid_motion_mlp.mlp_depth_plus', 1),
                      ('canonical_mlp.last_linear_scale', 2), ('non_rigid_motion_mlp.i_embed', -1),
                      ('canonical_mlp.time_input', True), ('non_rigid_motion_mlp.time_input', True),
                      ('rgb_history.last_num', 3), ('non_rigid_motion_mlp.multihead.enable', True),
                      ('canonical_mlp.multihead.enable', True), ('canonical_mlp.view_dir', True),
                      ('non_rigid_motion_model', 'transformer_encoder'), ('posevec.type', 'matrix'),
                      ('condition_code.type', 'local'),
                      ('embedder.module', 'core.nets.human_nerf.embedders.vocab_embedder'),
                      ('non_rigid_embedder.module', 'core.nets.human_nerf.embedders.fourier')]:
        c = get_cfg_defaults()
        node = c
        parts = path.split('.')
        for p_ in parts[:-1]:
            if p_ not in node:
                node[p_] = {}
            node = node[p_]
        node[parts[-1]] = val
        with pytest.raises(NotImplementedError, match=parts[-1]):
            check_config_is_built(c)
    # the reference's default module paths pass (compared by last component)
    c = get_cfg_defaults()
    c.embedder = {'module': 'core.nets.human_nerf.embedders.fourier'}
    c.non_rigid_embedder = {'module': 'core.nets.human_nerf.embedders.hannw_fourier'}
    check_config_is_built(c)


def test_trainer_states_mse_only_objective():
    """lossweights.lpips > 0 without an lpips_fn must not pass silently."""
    import warnings
    from humannerf_amd.network import Network
    from humannerf_amd.train import Trainer
    net = Network()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        tr = Trainer(net)
    assert any('lpips' in str(x.message) for x in w) and tr.objective == '0.2*mse'


def test_patch_sampler_equals_reference_sampler_draw_for_draw():
    """scene.PatchSampler (per-frame constants cached) vs scene.sample_patch_rays (pinned bit for bit to the
    reference's sampler by tests/golden/patches_s96.npz): same generator calls, same outputs."""
    from humannerf_amd import scene
    H, W = 96, 80
    yy, xx = np.mgrid[0:H, 0:W]
    bbox = (yy > 10) & (yy < 90) & (xx > 5) & (xx < 70)
    subj = (yy - 50) ** 2 / 900 + (xx - 40) ** 2 / 300 < 1
    ray_mask = bbox.reshape(-1)
    ps = scene.PatchSampler(ray_mask, subj, bbox, H, W)
    for seed in range(6):
        np.random.seed(seed)
        a = scene.sample_patch_rays(ray_mask, subj, bbox, 6, 32, H, W, subject_ratio=0.7)
        state_a = np.random.get_state()[1].copy()
        np.random.seed(seed)
        b = ps.draw(6, 32, subject_ratio=0.7)
        assert np.array_equal(np.random.get_state()[1], state_a)          # the generator advanced identically
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
        for k in a[1]:
            assert np.array_equal(a[1][k], b[1][k]), k


def test_f16_range_guard_schedule():
    """cfg.amd.f16_range_guard (the guard costs 3 % of a frame): 'audit' guards every ray chunk of the first frame
    rendered with a set of weights, then one chunk per frame, rotating; a weight change starts over; 'full' / 'off'
    guard everything / nothing; the exact fp32 mode has nothing to guard.  The mode strings are what ops hands to
    the C ABI (HNRF_MLP_NO_RANGE_GUARD / HNRF_MLP_GUARD_ONE_CHUNK, include/hnrf.h)."""
    from humannerf_amd import ops
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    net = Network()
    old = cfg.amd.get('f16_range_guard', 'audit')
    try:
        cfg.amd.f16_range_guard = 'audit'
        net._cnl_pack = (('f16x3', 1), None)
        assert net._guard_plan('f16x3', 8) == ('f16x3', None)
        got = [net._guard_plan('f16x3', 8) for _ in range(9)]
        assert [g[0] for g in got] == ['f16x3+guard1:%d' % (k % 8) for k in range(9)]
        assert [g[1] for g in got] == [{k % 8} for k in range(9)]
        net._cnl_pack = (('f16x3', 2), None)                                  # an optimizer step / a new checkpoint
        assert net._guard_plan('f16x3', 8) == ('f16x3', None)
        assert net._guard_plan('f16x3', 1) == ('f16x3+guard1:0', {0})         # one-chunk frames stay fully guarded
        assert net._guard_plan('f32', 8) == ('f32', None)
        cfg.amd.f16_range_guard = 'off'
        assert net._guard_plan('f16x3', 8) == ('f16x3+noguard', set())
        cfg.amd.f16_range_guard = 'full'
        assert net._guard_plan('f16x3', 8) == ('f16x3', None)
    finally:
        cfg.amd.f16_range_guard = old
    assert ops._mode_arg('f16x3') == 1 and ops._mode_arg('f32') == 0
    assert ops._mode_arg('f16x3+noguard') == 0x101 and ops._mode_arg('f16x3+guard1:5') == 0x50201


def test_point_deconv_keeps_the_reference_checkpoint_layout():
    """network._PointDeconv: parameter = central taps, state_dict / optimizer checkpoints = the reference's full tensors.
    The decoder's output equals F.conv_transpose3d on the full weight; the gradient of the compact parameter is the
    central block of the full gradient and the rest of the full gradient is exactly zero (the premise); a round trip
    through state_dict / GroupedAdam.state_dict changes nothing; Adam on the compact parameter moves the central taps
    exactly as Adam on the full tensor does and leaves the other 56 taps where they were."""
    import torch.nn.functional as F
    from humannerf_amd.network import MotionWeightVolumeDecoder, expand_central_taps, full_gradient
    from humannerf_amd.train import GroupedAdam
    torch.manual_seed(3)
    dec = MotionWeightVolumeDecoder(embedding_size=16, volume_size=8, total_bones=4)
    first = dec.decoder.block_conv[0]
    sd = dec.state_dict()
    kw = 'decoder.block_conv.0.weight'
    assert tuple(sd[kw].shape) == (1024, 512, 4, 4, 4) and tuple(first.weight.shape) == (1024, 512, 2, 2, 2)
    assert 'decoder.block_conv.0.weight_rest' not in sd
    pri = torch.rand(1, 5, 8, 8, 8) + 0.1
    pri = pri / pri.sum(1, keepdim=True)
    gout = torch.randn(1, 5, 8, 8, 8)
    out = dec(motion_weights_priors=pri)
    (out * gout).sum().backward()
    # reference evaluation on the FULL tensors with torch's own transposed convolution
    full = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    h = F.leaky_relu(F.linear(full['const_embedding'][None], full['decoder.block_mlp.0.weight'], full['decoder.block_mlp.0.bias']), 0.2)
    h = h.view(-1, 1024, 1, 1, 1)
    convs = sorted({int(k.split('.')[2]) for k in sd if k.startswith('decoder.block_conv.')})
    for i in convs:
        h = F.conv_transpose3d(h, full['decoder.block_conv.%d.weight' % i], full['decoder.block_conv.%d.bias' % i], stride=2, padding=1)
        if i != convs[-1]:
            h = F.leaky_relu(h, 0.2)
    ref = F.softmax(h + torch.log(pri), dim=1)
    assert (out - ref).abs().max() < 1e-6
    (ref * gout).sum().backward()
    gfull = full[kw].grad
    outer = gfull.clone()
    outer[:, :, 1:3, 1:3, 1:3] = 0
    assert float(outer.abs().max()) == 0.0                                  # 56 of 64 taps: exactly zero, always
    assert (full_gradient(first.weight) - gfull).abs().max() <= 1e-6 * float(gfull.abs().max())
    # one Adam step on both sides, from the SAME gradient (Adam's first step is lr g / (|g| + eps): last-bit differences
    # between two backward implementations would show wherever |g| ~ eps)
    wfull = sd[kw].clone().requires_grad_(True)
    wfull.grad = expand_central_taps(first.weight.grad)
    torch.optim.Adam([wfull], lr=1e-2).step()
    opt = GroupedAdam([{'params': [p]} for p in dec.parameters()], lr=1e-2)
    if not torch.cuda.is_available():
        for g in opt.param_groups:                                           # the fused launch needs a GPU: torch's loop here
            g['fused'] = False
        torch.optim.Adam.step(opt)
    else:
        opt.step()
    after = dec.state_dict()
    assert (after[kw] - wfull.detach()).abs().max() <= 1e-7
    rest = after[kw].clone()
    rest[:, :, 1:3, 1:3, 1:3] = sd[kw][:, :, 1:3, 1:3, 1:3]
    assert torch.equal(rest, sd[kw])                                         # untouched taps: bit for bit
    # checkpoints: moments in the reference's shape, zero outside the central taps; loading them back is the identity
    osd = opt.state_dict()
    idx = [n for n, _ in dec.named_parameters()].index(kw)
    m = osd['state'][idx]['exp_avg']
    assert tuple(m.shape) == (1024, 512, 4, 4, 4) and torch.equal(m, expand_central_taps(opt.state[first.weight]['exp_avg']))
    opt2 = GroupedAdam([{'params': [p]} for p in dec.parameters()], lr=1e-2)
    opt2.load_state_dict(osd)
    assert torch.equal(opt2.state[first.weight]['exp_avg'], opt.state[first.weight]['exp_avg'])
    dec2 = MotionWeightVolumeDecoder(embedding_size=16, volume_size=8, total_bones=4)
    dec2.load_state_dict(after)
    assert all(torch.equal(a, b) for a, b in zip(dec2.state_dict().values(), after.values()))
