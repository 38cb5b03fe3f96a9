"""Training path on the GPU: gradients from the HIP forward/backward (autograd.RenderRays)
vs the reference's gradients for the same scalar loss (tests/golden/grad_s64.npz), plus
kernel-level checks of the backward kernels against torch.autograd through the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_grad_oracle import compare_grads, grad_frame, reference_loss

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def exact_gradients(seeded_params, golden_dir):
    """fp64 torch.autograd through the oracle for the gradient fixture's frame and loss (CPU, ~20 s)."""
    from tests.test_grad_oracle import oracle_gradients
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)['grad_s64']
    g = np.load(os.path.join(golden_dir, 'grad_s64.npz'))
    return oracle_gradients(seeded_params, meta, g, torch.float64)[0]


@pytest.mark.parametrize('train_mode,operands', [('f32', 'f32'), ('f16x3', 'f32'), ('f16x3', 'f16')])
def test_network_gradients_match_reference(train_mode, operands, seeded_params, golden_dir, exact_gradients):
    """Gradients of all 55 parameter tensors, with the training kernels in exact-fp32 MFMA arithmetic and in the
    default split-f16 arithmetic,
      * vs the REFERENCE's own loss.backward() (tests/golden/grad_s64.npz) at the reference's noise floor: its fp32
        gradients are up to 4.4e-3 (norm) away from an fp64 evaluation (tests/test_grad_oracle.py), so 6e-3 / cosine
        0.99998 is what can be asked against it;
      * vs that fp64 evaluation (torch.autograd through the oracle): 6e-3 / 0.99998 as well (1e-2 / 0.99997 for the pose
        decoder, which sits behind Rodrigues, the kinematic chain and the 2^9 positional-encoding band).  This problem's
        gradient is not a smooth function of the forward's last bits: profiles/tools/pose_grad_noise.py evaluates the SAME op
        sequence (torch GEMVs for the pose MLP, exact-fp32 kernels) twice, the second time with dst_posevec multiplied by
        1 + 1.2e-7, and the gradients move by 7.8e-3 (pose decoder), 4.5e-3 (canonical MLP, layer 0 bias), 3.2e-3
        (non-rigid MLP) -- one discrete decision (a ReLU / voxel cell / clamp of one of the 4 096 samples) flips.  Replacing
        torch's GEMV by hnrf_pose_mlp_fwd (another summation order) flips the same one.  Typical measured values: worst
        tensor 7e-4 in norm when no such decision flips relative to the fp64 run, 6e-3 when one does; the split-f16
        arithmetic costs nothing measurable either way (the printed numbers say so).  What pins the kernels' arithmetic
        tightly are the per-kernel tests below (2e-5 against fp64 autograd of the same MLP / chain / warp)."""
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    from tests.test_grad_oracle import compare_exact
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)['grad_s64']
    g = np.load(os.path.join(golden_dir, 'grad_s64.npz'))
    fr = grad_frame(meta)
    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(dev()).train()
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev()) for k in keys}
    cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = meta['N_samples'], 0.0, False
    cfg.amd.train_mlp_mode = cfg.amd.train_dw_mode = cfg.amd.train_chain_mode = train_mode
    cfg.amd.train_operands = operands
    try:
        out = net(**data, iter_val=meta['iter_val'])
        assert len(out) == 11 and not out['weights_on_rays'].requires_grad       # network.py:776-789: all keys in train mode too
        loss = reference_loss(out, torch.from_numpy(g['loss_weights']).to(dev()))
        loss.backward()
    finally:
        cfg.N_samples, cfg.perturb = 128, 1.0
        cfg.amd.train_mlp_mode = cfg.amd.train_dw_mode = cfg.amd.train_chain_mode = 'f16x3'
        cfg.amd.train_operands = 'f16'
    assert abs(float(loss) - meta['loss']) <= 2e-4 * max(1.0, abs(meta['loss']))
    from humannerf_amd.network import full_gradient            # (the decoder's first layer trains its central taps only)
    grads = {k: (full_gradient(p).cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
             for k, p in net.named_parameters()}
    vs_ref = compare_grads(grads, g, rel_norm=6e-3, cos_min=0.99998)
    vs_exact = compare_exact(grads, exact_gradients, rel_norm=6e-3, cos_min=0.99998, loose=('pose_decoder.', 1e-2, 0.99997))
    print('gradients, training arithmetic', train_mode, 'operands', operands, '| vs reference', vs_ref, '| vs fp64', vs_exact)


def test_f16_operand_gradients_equal_fp32_operand_gradients(seeded_params, golden_dir):
    """the end-to-end bound above (6e-3, the problem's own noise) could not tell the default f16-operand
    weight-gradient path from one ten times worse.  This does: the SAME process runs the step twice on the same inputs,
    cfg.amd.train_operands = 'f16' (activations / dZ stored as f16, one f16 MFMA per product) and 'f32' (fp32 storage,
    22-bit split operands).  Forward and dX chains are the same kernels with the same bits in both runs, so no ReLU or
    voxel decision can flip between them: what differs is the operands' 11-bit rounding inside sums over 4 096
    samples.  Every one of the 55 tensors must agree to 1e-3 in norm (measured: printed)."""
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)['grad_s64']
    g = np.load(os.path.join(golden_dir, 'grad_s64.npz'))
    fr = grad_frame(meta)
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev()) for k in keys}
    lw = torch.from_numpy(g['loss_weights']).to(dev())
    cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = meta['N_samples'], 0.0, False
    grads = {}
    try:
        for operands in ('f16', 'f32'):
            cfg.amd.train_operands = operands
            net = Network()
            net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
            net = net.to(dev()).train()
            out = net(**data, iter_val=meta['iter_val'])
            reference_loss(out, lw).backward()
            grads[operands] = {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}
            grads[operands + '_rgb'] = out['rgb'].detach().cpu()
    finally:
        cfg.N_samples, cfg.perturb, cfg.amd.train_operands = 128, 1.0, 'f16'
    assert torch.equal(grads['f16_rgb'], grads['f32_rgb'])                   # same forward bits
    assert set(grads['f16']) == set(grads['f32']) and len(grads['f16']) == 55
    worst = (0.0, '')
    for k, a in grads['f16'].items():
        b = grads['f32'][k]
        e = float((a - b).norm() / b.norm())
        worst = max(worst, (e, k))
        assert e <= 1e-3, (k, e)
    print('f16 vs fp32 weight-gradient operands, end to end: worst tensor %.2e (%s)' % worst)


def test_composite_bwd_kernel():
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(5)
    R, S = 19, 128
    raw = rs.randn(R, S, 4).astype(np.float32) * 2
    raw[..., 3] = rs.randn(R, S) * 20 + 5
    mask = rs.uniform(0, 1.1, (R, S)).astype(np.float32)
    z = np.sort(1 + rs.uniform(0, 3, (R, S)).astype(np.float32), axis=1)
    rays_d = rs.randn(R, 3).astype(np.float32)
    bg = np.array([200., 100., 30.], dtype=np.float32)
    g_rgb, g_a, g_d = rs.randn(R, 3).astype(np.float32), rs.randn(R).astype(np.float32), rs.randn(R).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev())
    d_raw, d_mask = ops.composite_bwd(T(raw), T(mask), T(z), T(rays_d), T(bg), T(g_rgb), T(g_a), T(g_d))
    rt = torch.from_numpy(raw).double().requires_grad_(True)
    mt = torch.from_numpy(mask).double().requires_grad_(True)
    o = oracle.raw2outputs(rt, mt, torch.from_numpy(z).double(), torch.from_numpy(rays_d).double(),
                           torch.zeros(R, S, 3).double(), torch.from_numpy(bg).double())
    loss = (o['rgb'] * torch.from_numpy(g_rgb)).sum() + (o['alpha'] * torch.from_numpy(g_a)).sum() \
        + (o['depth'] * torch.from_numpy(g_d)).sum()
    loss.backward()
    for got, ref in ((d_raw, rt.grad), (d_mask, mt.grad)):
        err = (got.cpu().double() - ref).abs().max() / max(1.0, float(ref.abs().max()))
        assert err <= 2e-5, float(err)


# small: global atomics; large: LDS-privatised grid; S = 50: a wave's lanes span several rays and the last wave is
# ragged; span 0.05: a ray stays in one voxel cell for tens of samples (long runs for the atomic fold of K1')
@pytest.mark.parametrize('R,G,S,span', [(23, 16, 64, 0.5), (1100, 32, 64, 0.5), (37, 32, 50, 0.5), (1400, 32, 50, 0.5),
                                        (1100, 32, 128, 0.05)])
def test_sample_warp_bwd_kernel(R, G, S, span):
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(9)
    B = 24
    rays_o = rs.uniform(-0.3, 0.3, (R, 3)).astype(np.float32)
    rays_d = rs.uniform(-1, 1, (R, 3)).astype(np.float32)
    near = rs.uniform(0.0, 0.2, (R, 1)).astype(np.float32)
    far = near + rs.uniform(span, 2 * span, (R, 1)).astype(np.float32)
    Rs = (np.eye(3)[None] + 0.1 * rs.randn(B, 3, 3)).astype(np.float32)
    Ts = (0.1 * rs.randn(B, 3)).astype(np.float32)
    vol = rs.uniform(0, 0.1, (B + 1, G, G, G)).astype(np.float32)
    bmin = np.array([-1.0, -1.1, -0.9], dtype=np.float32)
    bscale = (2.0 / np.array([2.0, 2.2, 1.8])).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev())
    z, xs, mask, _ = ops.sample_warp(T(rays_o), T(rays_d), T(near), T(far), None, T(Rs), T(Ts), T(vol), T(bmin),
                                     T(bscale), S)
    gx = rs.randn(R, S, 3).astype(np.float32)
    gm = rs.randn(R, S).astype(np.float32)
    d_vol, d_Rs, d_Ts = ops.sample_warp_bwd(T(rays_o), T(rays_d), z, T(Rs), T(Ts), T(vol), T(bmin), T(bscale), xs,
                                            mask, T(gx), T(gm))
    # oracle autograd in fp64
    Rt = torch.from_numpy(Rs).double().requires_grad_(True)
    Tt = torch.from_numpy(Ts).double().requires_grad_(True)
    vt = torch.from_numpy(vol).double().requires_grad_(True)
    zo = oracle.z_values(torch.from_numpy(near).double(), torch.from_numpy(far).double(), S)
    pts = torch.from_numpy(rays_o).double()[:, None] + torch.from_numpy(rays_d).double()[:, None] * zo[:, :, None]
    xo, mo, _ = oracle.sample_motion_fields(pts.reshape(-1, 3), Rt, Tt, vt, torch.from_numpy(bmin).double(),
                                            torch.from_numpy(bscale).double())
    loss = (xo * torch.from_numpy(gx).double().reshape(-1, 3)).sum() + (mo * torch.from_numpy(gm).double().reshape(-1)).sum()
    loss.backward()
    # d w / d pos of a trilinear lookup jumps at voxel faces: a sample whose fp32 position falls on the
    # other side of a face than its fp64 position flips one term of the 1.7 M-term motion-base sums
    # (~1e-4 of the samples sit within fp32 rounding of a face), so those reductions agree to ~1 %
    # only; the volume gradient (no spatial derivative) agrees to 2e-3
    tol = {'vol': 2e-3, 'Rs': 2e-3 if R < 100 else 3e-2, 'Ts': 2e-3 if R < 100 else 3e-2}
    for name, got, ref in (('vol', d_vol, vt.grad), ('Rs', d_Rs, Rt.grad), ('Ts', d_Ts, Tt.grad)):
        err = (got.cpu().double() - ref).abs().max() / max(1e-6, float(ref.abs().max()))
        assert err <= tol[name], (name, float(err))
    assert float(d_vol[-1].abs().max()) == 0       # background channel never sampled


def test_pe_bwd_kernel():
    from humannerf_amd import ops
    from oracle import oracle
    rs = np.random.RandomState(2)
    P = 777
    x = rs.uniform(-1.2, 1.2, (P, 3)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev())
    for nb, inc, hw in ((10, True, None), (6, False, np.array([1, 1, 0.7, 0.2, 0, 0], dtype=np.float32))):
        C = (3 if inc else 0) + 6 * nb
        g = rs.randn(P, C).astype(np.float32)
        got = ops.pe_bwd(T(x), T(g), None if hw is None else T(hw), nb, inc)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        pe = oracle.fourier_pe(xt, nb) if inc else oracle.hann_pe(xt, torch.from_numpy(hw).double())
        (pe * torch.from_numpy(g).double()).sum().backward()
        err = (got.cpu().double() - xt.grad).abs().max() / float(xt.grad.abs().max())
        assert err <= 1e-4, float(err)


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
def test_training_forward_equals_inference_forward(mode):
    """The activation-saving forward returns the same raw values as the inference kernel, and the
    saved activations are what the oracle computes."""
    from humannerf_amd import ops
    from oracle import oracle
    from tests.test_gpu_parity import _mlp_states
    rs = np.random.RandomState(4)
    st = _mlp_states(rs)
    P = 333
    xyz = rs.uniform(-1.2, 1.2, (P, 3)).astype(np.float32)
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    T = lambda a: torch.from_numpy(a).to(dev())
    ws = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
    bs = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
    packed = ops.canonical_pack(ws, bs, mode)
    raw = ops.canonical(T(xyz), packed, mode)
    raw_t, pe, acts, bits = ops.canonical_train(T(xyz), packed, mode)
    assert bits.shape == (8, P, 8)
    assert torch.equal(raw, raw_t)
    pe_ref = oracle.fourier_pe(torch.from_numpy(xyz), 10)
    assert (pe.cpu() - pe_ref).abs().max() <= 1e-6
    h = torch.relu(torch.nn.functional.linear(pe_ref, torch.from_numpy(st['cnl_mlp.module.pts_linears.0.weight']),
                                              torch.from_numpy(st['cnl_mlp.module.pts_linears.0.bias'])))
    assert (acts[0].cpu() - h).abs().max() <= 1e-5
    # every layer's saved activations against an fp64 evaluation of the same MLP
    names = [f'cnl_mlp.module.pts_linears.{i}' for i in idx]
    pe64 = pe_ref.double()
    h64 = pe64
    for l, n in enumerate(names):
        if l == 5:
            h64 = torch.cat([pe64, h64], -1)
        h64 = torch.relu(torch.nn.functional.linear(h64, torch.from_numpy(st[n + '.weight']).double(),
                                                    torch.from_numpy(st[n + '.bias']).double()))
        assert (acts[l].double().cpu() - h64).abs().max() <= 2e-5 * max(1.0, float(h64.abs().max())), l


def _rows(m, P, blocked):
    """first P rows of an activation matrix, undoing the blocked layout of the f16-operand mode"""
    if not blocked:
        return m[:P]
    P128, W = m.shape
    m5 = m.view(P128 // 32, W // 32, 8, 32, 4)                                                   # (block, tile, k, slot, j)
    k = torch.arange(8, device=m.device).view(8, 1)
    src = (torch.arange(32, device=m.device).view(1, 32) ^ (4 * k))                              # sample c sits in slot c ^ 4 k
    m5 = torch.gather(m5, 3, src.view(1, 1, 8, 32, 1).expand(m5.shape[0], m5.shape[1], 8, 32, 4))
    return m5.permute(0, 3, 1, 2, 4).contiguous().view(P128, W)[:P]


def _torch_mlp(x_in, ws, bs, skip_layer, skip_order, pe_fn, masks):
    """Plain torch restatement of mlp_rgb_sigma.py:132-198 / mlp_offset.py forward on a PE function.
    relu(z) is applied as z * masks[l] with the sign pattern of the GPU forward, so that a pre-activation
    within rounding of zero cannot make the two gradients differ by a whole term."""
    pe = pe_fn(x_in)
    h = pe
    for l in range(len(ws) - 1):
        if l == skip_layer:
            h = torch.cat([pe, h], -1) if skip_order == 'pe_first' else torch.cat([h, pe], -1)
        h = torch.nn.functional.linear(h, ws[l], bs[l]) * masks[l]
    return torch.nn.functional.linear(h, ws[-1], bs[-1])


@pytest.mark.parametrize('mode', ['f32', 'f16x3', 'f16x3h'])
def test_canonical_backward_chain_and_weight_gradients_match_autograd(mode):
    """hnrf_canonical_bwd (dX chain + fused PE') and hnrf_mlp_dw against torch.autograd of the same MLP
    (fp64 on the CPU): dZ of every layer, d_xyz, and every dW / db."""
    from humannerf_amd import ops
    from humannerf_amd.autograd import _weight_grads
    from oracle import oracle
    from tests.test_gpu_parity import _mlp_states
    rs = np.random.RandomState(11)
    st = _mlp_states(rs)
    P = 777                                                    # ragged: not a multiple of 32 / 128
    xyz = rs.uniform(-1.2, 1.2, (P, 3)).astype(np.float32)
    g_raw = (rs.standard_normal((P, 4)) * (rs.uniform(size=(P, 1)) > 0.3)).astype(np.float32)
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    names = [f'cnl_mlp.module.pts_linears.{i}' for i in idx] + ['cnl_mlp.module.output_linear.0']
    T = lambda a: torch.from_numpy(a).to(dev())
    ws, bs = [T(st[n + '.weight']) for n in names], [T(st[n + '.bias']) for n in names]
    half = mode == 'f16x3h'
    raw, pe, acts, bits = ops.canonical_train(T(xyz), ops.canonical_pack(ws, bs, 'f16x3' if half else mode), mode)
    dZ, d_xyz, amax = ops.canonical_bwd(T(xyz), T(g_raw), bits, ws, mode)
    if half:
        from humannerf_amd.autograd import _weight_grads_h
        assert dZ.dtype == torch.float16 and acts.dtype == torch.float16 and pe.shape == (P, 64) and amax.shape == (8,)
        assert dZ.shape == (8, (P + 127) // 128 * 128, 256) and acts.shape == dZ.shape
        assert float(pe[:, 63].abs().max()) == 0.0
        gW, gb = _weight_grads_h(dZ, amax, acts, pe, T(g_raw), ws, skip_layer=5, skip_order='pe_first', npe=63)
    else:
        assert torch.equal(amax.amax(1), dZ.abs().amax(dim=(1, 2)))
        gW, gb = _weight_grads(dZ, acts, pe, T(g_raw), ws, skip_layer=5, skip_order='pe_first', amax=amax, mode=mode)

    x64 = torch.from_numpy(xyz).double().requires_grad_(True)
    w64 = [torch.from_numpy(st[n + '.weight']).double().requires_grad_(True) for n in names]
    b64 = [torch.from_numpy(st[n + '.bias']).double().requires_grad_(True) for n in names]
    masks = [(_rows(acts[l], P, half) > 0).double().cpu() for l in range(8)]
    out = _torch_mlp(x64, w64, b64, 5, 'pe_first', lambda x: oracle.fourier_pe(x, 10), masks)
    out.backward(torch.from_numpy(g_raw).double())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    errs = {'d_xyz': rel(d_xyz, x64.grad)}
    errs.update({'W%d' % l: rel(gW[l], w64[l].grad) for l in range(9)})
    errs.update({'b%d' % l: rel(gb[l], b64[l].grad) for l in range(9)})
    print('canonical backward', mode, 'worst rel err %.2e (%s)' % max((v, k) for k, v in errs.items()))
    # f16 operands: every product of a weight-gradient sum carries two 11-bit roundings (rms 2.9e-4 of the product).  The
    # terms of THIS test have random signs, so the sum is a random walk and its error stays ~3e-4 of its own size
    # (measured 3.7e-4 of the largest element); for the coherent part of a real gradient it falls with 1 / sqrt(samples).
    # End to end (test_network_gradients_match_reference) the f16 operands change nothing that can be measured against
    # fp64.  d_xyz comes from the 22-bit chain in every mode
    assert errs['d_xyz'] <= 2e-5
    for k, v in errs.items():
        assert v <= (1e-3 if half else 2e-5), (k, v)
    if half:      # the per-layer scales of the stored dZ are powers of two
        assert float(amax.min()) > 0 and all(float(torch.log2(a)) == round(float(torch.log2(a))) for a in amax)


@pytest.mark.parametrize('regime', ['scaled', 'fresh_init', 'tiny_hidden'])
@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
def test_nonrigid_backward_chain_and_weight_gradients_match_autograd(mode, regime):
    """hnrf_nonrigid_bwd + hnrf_mlp_dw against torch.autograd (fp64): d_x_skel includes the identity path of
    xyz = x_skel + offset, the Hann window weights scale the PE gradient, the condition code enters layer 0.
    Regimes: see tests/test_gpu_parity.py::_apply_regime -- ``fresh_init`` is the state every training run starts in
    (gradients reach layers 0..10 through a last layer of +-1e-5), each tensor is compared relative to its OWN
    magnitude."""
    from humannerf_amd import ops
    from humannerf_amd.autograd import _weight_grads
    from oracle import oracle
    from tests.test_gpu_parity import _mlp_states, _apply_regime
    rs = np.random.RandomState(12)
    st = _apply_regime(_mlp_states(rs), regime, rs)
    P = 1000
    x = rs.uniform(-1.0, 1.0, (P, 3)).astype(np.float32)
    g_xyz = rs.standard_normal((P, 3)).astype(np.float32)
    cond = (rs.standard_normal(69) * 0.3).astype(np.float32)
    hann = np.array([1.0, 1.0, 0.75, 0.25, 0.0, 0.0], np.float32)
    names = [f'non_rigid_mlp.module.block_mlps.{i}' for i in (0, 2, 4, 6, 8, 10, 12)]
    T = lambda a: torch.from_numpy(a).to(dev())
    ws, bs = [T(st[n + '.weight']) for n in names], [T(st[n + '.bias']) for n in names]
    xyz, off, pe, acts, bits = ops.nonrigid_train(T(x), T(hann), ops.nonrigid_pack(ws, bs, T(cond), mode), mode)
    dZ, d_x, amax = ops.nonrigid_bwd(T(x), T(hann), T(g_xyz), bits, ws, mode)
    assert torch.equal(amax.amax(1), dZ.abs().amax(dim=(1, 2)))
    gW, gb = _weight_grads(dZ, acts, pe, T(g_xyz), ws, skip_layer=4, skip_order='h_first', amax=amax, mode=mode)
    gW[0] = torch.cat([gb[0][:, None] * T(cond).reshape(1, -1), gW[0]], dim=1)

    x64 = torch.from_numpy(x).double().requires_grad_(True)
    w64 = [torch.from_numpy(st[n + '.weight']).double().requires_grad_(True) for n in names]
    b64 = [torch.from_numpy(st[n + '.bias']).double().requires_grad_(True) for n in names]
    c64, h64 = torch.from_numpy(cond).double(), torch.from_numpy(hann).double()

    def pe_fn(xx):
        bands = []
        for k in range(6):
            bands += [h64[k] * torch.sin(xx * 2.0 ** k), h64[k] * torch.cos(xx * 2.0 ** k)]
        return torch.cat(bands, -1)
    pe64 = pe_fn(x64)
    h = torch.cat([c64.expand(P, 69), pe64], -1)
    for l in range(6):
        if l == 4:
            h = torch.cat([h, pe64], -1)
        h = torch.nn.functional.linear(h, w64[l], b64[l]) * (acts[l] > 0).double().cpu()
    out = x64 + torch.nn.functional.linear(h, w64[6], b64[6])
    out.backward(torch.from_numpy(g_xyz).double())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert (xyz.double().cpu() - out.detach()).abs().max() <= 1e-5
    from humannerf_amd.autograd import OperandRangeGuard
    flags = [bool(f) for f in OperandRangeGuard.flags([acts], T(g_xyz))]
    assert flags == [regime == 'tiny_hidden', False, False]     # the guard sees the layer whose activations are ~1e-6
    errs = {'d_x': rel(d_x, x64.grad)}
    errs.update({'W%d' % l: rel(gW[l], w64[l].grad) for l in range(7)})
    errs.update({'b%d' % l: rel(gb[l], b64[l].grad) for l in range(7)})
    print('nonrigid backward', mode, regime, 'worst rel err %.2e (%s)' % max((v, k) for k, v in errs.items()),
          'max|dW0| %.2e max|dW6| %.2e' % (float(w64[0].grad.abs().max()), float(w64[6].grad.abs().max())))
    for k, v in errs.items():
        # split-f16 with a hidden layer whose activations are ~1e-6: the weight gradient of the NEXT layer multiplies
        # operands below f16's normal range (measured 2.4e-3) -- outside the arithmetic's premise, which is why
        # OperandRangeGuard (checked above) raises in training and points to the exact 'f32' kernels
        # (round 3: 2.4e-3 -> 1.3e-2 measured, since the forward images keep the weights' low parts un-lifted -- a layer of
        # 1e-6 weights then lives on f16 subnormals alone; its output is 1e-6 of the next layer's bias path)
        lim = 2e-2 if (mode != 'f32' and regime == 'tiny_hidden' and k == 'W3') else 2e-5
        assert v <= lim, (k, v, errs)


@pytest.mark.parametrize('variant', ['tpose', 'early_iter', 'stratified'])
def test_network_gradients_match_oracle_autograd_in_other_branches(seeded_params, variant):
    """Branches of Network.forward the reference-gradient fixture does not visit, against torch.autograd through the
    CPU oracle (itself pinned to the reference's gradients by tests/test_grad_oracle.py):
    tpose = cfg.ignore_non_rigid_motions (network.py:264-277: no non-rigid MLP, its parameters get no gradient);
    early_iter = iter_val below the pose-decoder / non-rigid kick-in (condition code zeroed, Hann window closed);
    stratified = perturb > 0 with injected uniforms."""
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network, full_gradient
    from oracle import oracle
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, ray_stride=29)       # ~270 rays x 64 samples
    R = fr['rays'].shape[1]
    rs = np.random.RandomState(3)
    lw = rs.standard_normal((R, 5)).astype(np.float32)
    t_rand = rs.uniform(size=(R, 64)).astype(np.float32) if variant == 'stratified' else None
    iter_val = 100.0 if variant == 'early_iter' else 1e7
    kw = dict(iter_val=iter_val, N_samples=64)
    if variant == 'tpose':
        kw['ignore_non_rigid_motions'] = True
    if t_rand is not None:
        kw['t_rand'] = torch.from_numpy(t_rand)
    # fp64 oracle as the comparator, and a few hundred rays (with ~25 rays the evaluation noise of a single
    # ReLU / voxel-cell decision was ~2 % of a tensor's gradient and the bound had to be 5e-2: a 5 % error in one of
    # these branches would have passed).  Bounds in norm: 2e-2 (measured worst 1.2e-2: a non-rigid layer in the stratified branch),
    # pose decoder 3e-2 (measured 2.0e-2 stratified, 1.0e-2 tpose; early_iter, where no ReLU sits at a kink: 1.9e-4); in
    # direction 1 - cos <= 5e-4 (measured 1.4e-4 on the first canonical layer in the tpose branch: single ReLU / cell
    # decisions of the ~17 000 samples still show in direction before they show in norm); the worst tensor is printed.
    state = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in seeded_params.items()}
    ref_out = oracle.render(state, fr, dtype=torch.float64, **kw)
    ref_loss = reference_loss(ref_out, lw)
    ref_loss.backward()

    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(dev()).train()
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev()) for k in keys}
    if t_rand is not None:
        data['t_rand'] = torch.from_numpy(t_rand).to(dev())
    cfg.N_samples, cfg.perturb = 64, (1.0 if t_rand is not None else 0.0)
    cfg.ignore_non_rigid_motions = variant == 'tpose'
    try:
        out = net(**data, iter_val=iter_val)
        loss = reference_loss(out, torch.from_numpy(lw).to(dev()))
        loss.backward()
    finally:
        cfg.N_samples, cfg.perturb, cfg.ignore_non_rigid_motions = 128, 1.0, False
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-4 * max(1.0, abs(float(ref_loss.detach())))
    checked, worst, bad = 0, (0.0, 0.0, ''), []
    for name, p in net.named_parameters():
        ref = state[name].grad
        got = full_gradient(p)
        if ref is None or float(ref.norm()) == 0.0:
            assert got is None or float(got.norm()) <= 1e-12, name
            continue
        g, r = got.double().cpu().reshape(-1), ref.double().reshape(-1)
        e_norm = abs(float(g.norm()) - float(r.norm())) / float(r.norm())
        e_cos = 1.0 - float(g @ r / (g.norm() * r.norm()))
        worst = max(worst, (e_norm, e_cos, name))
        lim_n, lim_c = (3e-2, 1e-3) if name.startswith('pose_decoder.') else (2e-2, 5e-4)
        if e_norm > lim_n or e_cos > lim_c:
            bad.append((name, e_norm, e_cos))
        checked += 1
    print('other branches', variant, 'rays', R, 'worst tensor: norm err %.2e, 1 - cos %.2e (%s)' % worst)
    assert not bad, bad
    assert checked >= (40 if variant != 'tpose' else 26)


def test_training_gradients_are_additive_over_ray_subsets_at_full_size(seeded_params):
    """Size-independent property at BASELINE's training size (6 144 rays x 128 samples = 786 k samples, where the
    oracle is too slow): for a loss that is a sum over rays, the gradient of the whole batch equals the sum of the
    gradients of its two halves -- through the split-f16 forward, both dX chains, the slice-reduced dW kernels, K1'
    and the decoder.  Also: running the same batch twice gives bit-identical gradients (no atomics on the MLP path
    whose order could matter ... the weight-volume / motion-base gradients use float atomics and are compared with
    a tolerance)."""
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    d = dev()
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
    idx = (np.arange(6144) * 37) % (512 * 512)
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(d) for k in keys}
    lw = torch.from_numpy(np.random.RandomState(7).standard_normal((6144, 3)).astype(np.float32)).to(d)
    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(d).train()

    def grads(sel):
        net.zero_grad(set_to_none=True)
        sub = dict(data, rays=data['rays'][:, idx[sel]].contiguous(), near=data['near'][idx[sel]].contiguous(),
                   far=data['far'][idx[sel]].contiguous())
        out = net(**sub, iter_val=1e7)
        (out['rgb'] * lw[sel]).sum().backward()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    cfg.N_samples, cfg.perturb = 128, 0.0
    try:
        whole = grads(np.arange(6144))
        again = grads(np.arange(6144))
        a, b = grads(np.arange(0, 3072)), grads(np.arange(3072, 6144))
    finally:
        cfg.perturb = 1.0
    assert len(whole) >= 50
    for k, g in whole.items():
        scale = float(g.abs().max()) + 1e-30
        mlp = 'cnl_mlp' in k or 'non_rigid_mlp' in k
        rep = float((g - again[k]).abs().max()) / scale
        assert rep == 0.0 if mlp else rep <= 1e-4, (k, rep)
        err = float((g - (a[k] + b[k])).abs().max()) / scale
        assert err <= 2e-4, (k, err)


def test_shared_pe_evaluation_is_bit_identical():
    """The split-f16 kernels let the two lanes of a sample share the sin/cos evaluations (hnrf_sincos.h,
    pe_sincos_shared); the fp32 kernels evaluate everything on every lane.  Same octave phases, same polynomials:
    the saved positional encodings must agree bit for bit (canonical: 10 bands; non-rigid: 6 bands x Hann weights)."""
    from humannerf_amd import ops
    from tests.test_gpu_parity import _mlp_states
    rs = np.random.RandomState(21)
    st = _mlp_states(rs)
    P = 4099
    x = rs.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    x[0] = [0.0, 1e-30, -1.0]                      # zero, a denormal-scale argument, an exact turn fraction
    T = lambda a: torch.from_numpy(a).to(dev())
    idx = [0, 2, 4, 6, 8, 10, 12, 14]
    cw = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
    cb = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
    pes = [ops.canonical_train(T(x), ops.canonical_pack(cw, cb, m), m)[1] for m in ('f32', 'f16x3')]
    assert torch.equal(pes[0], pes[1])
    names = [f'non_rigid_mlp.module.block_mlps.{i}' for i in (0, 2, 4, 6, 8, 10, 12)]
    nw, nb = [T(st[n + '.weight']) for n in names], [T(st[n + '.bias']) for n in names]
    hann = T(np.array([1.0, 1.0, 0.75, 0.25, 0.5, 0.0], np.float32))
    cond = T(np.zeros(69, np.float32))
    pen = [ops.nonrigid_train(T(x), hann, ops.nonrigid_pack(nw, nb, cond, m), m)[2] for m in ('f32', 'f16x3')]
    assert torch.equal(pen[0], pen[1])


@pytest.mark.parametrize('P,n_out,n_in', [(4096, 256, 256), (1000, 256, 256), (777, 128, 128), (3001, 256, 128),
                                           (2000, 128, 256), (5000, 256, 63), (1234, 128, 36), (50000, 256, 256)])
def test_half_operand_weight_gradient_kernel(P, n_out, n_in):
    """hnrf_mlp_dw_h (transposed LDS reads of row-major f16 operands) against an fp64 product of the SAME f16
    values -- exact up to fp32 accumulation -- on ragged sample counts, every built shape, with a dZ scale, ReLU-like
    sparsity and rows past the last slice."""
    from humannerf_amd import ops
    rs = np.random.RandomState(P + n_in)
    scale = 64.0
    dZ = (rs.standard_normal((P, n_out)) * (rs.uniform(size=(P, n_out)) > 0.5)).astype(np.float32)
    nip = 64 if n_in <= 64 else n_in
    X = np.zeros((P, nip), np.float32)
    X[:, :n_in] = np.maximum(rs.standard_normal((P, n_in)), 0) * 0.7
    dZh = torch.from_numpy(dZ * scale).to(dev()).half()
    Xh = torch.from_numpy(X).to(dev()).half()
    sc = torch.tensor([scale], device=dev())
    dW, db = ops.mlp_dw_h(dZh, Xh, dz_scale=sc, n_in=n_in)
    ref = (dZh.double().cpu().T @ Xh.double().cpu())[:, :n_in] / scale
    refb = dZh.double().cpu().sum(0) / scale
    e_w = float((dW.double().cpu() - ref).abs().max() / ref.abs().max())
    e_b = float((db.double().cpu() - refb).abs().max() / refb.abs().max())
    print('dw_h', P, n_out, n_in, 'rel err dW %.2e db %.2e' % (e_w, e_b))
    assert e_w <= 2e-6 and e_b <= 2e-6
    # a view into a wider weight (the column blocks of a skip layer) and no bias
    wide = torch.zeros(n_out, n_in + 70, device=dev())
    out, none = ops.mlp_dw_h(dZh, Xh, dW_out=wide[:, 70:], want_db=False, dz_scale=sc, n_in=n_in)
    assert none is None and torch.equal(wide[:, 70:], dW) and float(wide[:, :70].abs().max()) == 0.0
    # determinism
    dW2, _ = ops.mlp_dw_h(dZh, Xh, dz_scale=sc, n_in=n_in)
    assert torch.equal(dW, dW2)


def _to_blocked(m):
    """row-major (P, W) f16 -> the blocked layout of hnrf_mlp_dw_h, padded to a multiple of 128 samples (torch ops)."""
    P, W = m.shape
    P128 = (P + 127) // 128 * 128
    pad = torch.zeros(P128, W, dtype=m.dtype, device=m.device)
    pad[:P] = m
    # (block, c) x (tile, k = 2 g + h, j)  ->  (block, tile, k, slot = c ^ 4 k, j)
    m5 = pad.view(P128 // 32, 32, W // 32, 8, 4).permute(0, 2, 3, 1, 4).contiguous()          # (block, tile, k, c, j)
    k = torch.arange(8, device=m.device).view(8, 1)
    src = (torch.arange(32, device=m.device).view(1, 32) ^ (4 * k))                              # slot s holds sample s ^ 4 k
    m5 = torch.gather(m5, 3, src.view(1, 1, 8, 32, 1).expand(m5.shape[0], m5.shape[1], 8, 32, 4))
    return m5.view(P128, W)


@pytest.mark.parametrize('P,n_out,n_in', [(4096, 256, 256), (1000, 256, 256), (777, 128, 128), (3001, 256, 63), (50000, 256, 256)])
def test_half_operand_weight_gradient_kernel_blocked_layout(P, n_out, n_in):
    """The same with the operands in the blocked layout the training kernels write (dZ always, X for the hidden
    layers): bit-identical to the row-major form."""
    from humannerf_amd import ops
    rs = np.random.RandomState(P + n_in + 1)
    dZh = torch.from_numpy((rs.standard_normal((P, n_out)) * 8).astype(np.float32)).to(dev()).half()
    nip = 64 if n_in <= 64 else n_in
    X = np.zeros((P, nip), np.float32)
    X[:, :n_in] = np.maximum(rs.standard_normal((P, n_in)), 0)
    Xh = torch.from_numpy(X).to(dev()).half()
    sc = torch.tensor([8.0], device=dev())
    want_W, want_b = ops.mlp_dw_h(dZh, Xh, dz_scale=sc, n_in=n_in)
    xb = n_in >= 128
    got_W, got_b = ops.mlp_dw_h(_to_blocked(dZh), _to_blocked(Xh) if xb else Xh, dz_scale=sc, n_in=n_in, P=P,
                                z_blocked=True, x_blocked=xb)
    assert torch.equal(got_W, want_W) and torch.equal(got_b, want_b)
    if n_in in (128, 256):                 # heads read blocked activations
        for heads in (4, 3):               # (canonical head: 4 outputs; non-rigid head: 3)
            dY = torch.from_numpy(rs.standard_normal((P, heads)).astype(np.float32)).to(dev())
            a = ops.mlp_dw_h(dY, Xh)
            b = ops.mlp_dw_h(dY, _to_blocked(Xh), P=P, x_blocked=True)                   # (coalesced form: other summation order)
            assert a[0].shape == (heads, n_in) and b[0].shape == (heads, n_in)
            assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(a[0].abs().max())
            assert float((a[1] - b[1]).abs().max()) <= 2e-6 * float(a[1].abs().max()) + 1e-4
            ref = dY.double().T @ Xh.double()
            assert float((b[0].double() - ref).abs().max() / ref.abs().max()) <= 2e-6


def test_half_operand_head_gradient_kernel():
    from humannerf_amd import ops
    rs = np.random.RandomState(5)
    for P, n_out, n_in in ((3000, 4, 256), (1111, 3, 128)):
        dY = rs.standard_normal((P, n_out)).astype(np.float32)
        Xh = torch.from_numpy(np.maximum(rs.standard_normal((P, n_in)), 0).astype(np.float32)).to(dev()).half()
        dW, db = ops.mlp_dw_h(torch.from_numpy(dY).to(dev()), Xh)
        ref = torch.from_numpy(dY).double().T @ Xh.double().cpu()
        assert float((dW.double().cpu() - ref).abs().max() / ref.abs().max()) <= 2e-6
        assert float((db.double().cpu() - torch.from_numpy(dY).double().sum(0)).abs().max()) <= 1e-3


def test_motion_basis_kernels_match_fp64_autograd():
    """hnrf_motion_basis_fwd / _bwd (one single-wave kernel each) vs torch fp64 autograd through the oracle's
    MotionBasisComputer restatement (network_util.py:125-156): outputs and the gradients w.r.t. dst_Rs / dst_Ts."""
    from humannerf_amd import scene
    from humannerf_amd.network import motion_basis, motion_basis_torch
    from oracle import oracle
    for seed in (0, 1, 2):
        rs = np.random.RandomState(seed)
        poses = rs.randn(72) * 0.4
        dst_Rs, dst_Ts = scene.body_pose_to_body_RTs(poses, scene.TPOSE_JOINTS)
        gt = scene.get_canonical_global_tfms(scene.TPOSE_JOINTS)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
        R_in, T_in = T(dst_Rs).requires_grad_(True), T(dst_Ts).requires_grad_(True)
        Rs, Ts = motion_basis(R_in, T_in, T(gt))
        gR, gT = rs.randn(24, 3, 3).astype(np.float32), rs.randn(24, 3).astype(np.float32)
        ((Rs * T(gR)).sum() + (Ts * T(gT)).sum()).backward()
        R64 = torch.from_numpy(dst_Rs).double().requires_grad_(True)
        T64 = torch.from_numpy(dst_Ts).double().requires_grad_(True)
        Rr, Tr = oracle.motion_basis(R64, T64, torch.from_numpy(gt).double())
        ((Rr * torch.from_numpy(gR).double()).sum() + (Tr * torch.from_numpy(gT).double()).sum()).backward()
        assert float((Rs.detach().double().cpu() - Rr.detach()).abs().max()) <= 2e-7
        assert float((Ts.detach().double().cpu() - Tr.detach()).abs().max()) <= 2e-7 * max(1.0, float(Tr.abs().max()))
        for got, ref in ((R_in.grad, R64.grad), (T_in.grad, T64.grad)):
            assert float((got.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        # and the torch restatement it replaces agrees too (fp32)
        R2, T2 = motion_basis_torch(T(dst_Rs), T(dst_Ts), T(gt))
        assert float((R2 - Rs.detach()).abs().max()) <= 5e-6 and float((T2 - Ts.detach()).abs().max()) <= 5e-6


def test_refined_motion_basis_kernels_match_fp64_autograd():
    """hnrf_refined_motion_basis_fwd / _bwd: the pose refinement's Rodrigues correction (network_util.py:57-83,
    network.py:677-688) folded into the kinematics kernels, vs torch fp64 autograd through the oracle's rodrigues +
    motion_basis; rvec magnitudes from the freshly initialised refiner's 1e-5 up to large corrections."""
    from humannerf_amd import scene
    from humannerf_amd.network import motion_basis
    from oracle import oracle
    for seed, mag in ((0, 1e-5), (1, 1e-2), (2, 0.3), (3, 2.0)):
        rs = np.random.RandomState(seed)
        dst_Rs, dst_Ts = scene.body_pose_to_body_RTs(rs.randn(72) * 0.4, scene.TPOSE_JOINTS)
        gt = scene.get_canonical_global_tfms(scene.TPOSE_JOINTS)
        rv = (rs.randn(23, 3) * mag).astype(np.float32)
        rv[5] = 0.0                                                    # exactly zero correction: theta = sqrt(1e-5)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
        R_in, T_in, v_in = T(dst_Rs).requires_grad_(True), T(dst_Ts).requires_grad_(True), T(rv).requires_grad_(True)
        Rs, Ts = motion_basis(R_in, T_in, T(gt), v_in)
        gR, gT = rs.randn(24, 3, 3).astype(np.float32), rs.randn(24, 3).astype(np.float32)
        ((Rs * T(gR)).sum() + (Ts * T(gT)).sum()).backward()
        R64 = torch.from_numpy(dst_Rs).double().requires_grad_(True)
        T64 = torch.from_numpy(dst_Ts).double().requires_grad_(True)
        v64 = torch.from_numpy(rv).double().requires_grad_(True)
        Rc = torch.cat([R64[0:1], torch.matmul(R64[1:], oracle.rodrigues(v64))], dim=0)
        Rr, Tr = oracle.motion_basis(Rc, T64, torch.from_numpy(gt).double())
        ((Rr * torch.from_numpy(gR).double()).sum() + (Tr * torch.from_numpy(gT).double()).sum()).backward()
        assert float((Rs.detach().double().cpu() - Rr.detach()).abs().max()) <= 2e-7
        assert float((Ts.detach().double().cpu() - Tr.detach()).abs().max()) <= 2e-7 * max(1.0, float(Tr.abs().max()))
        for got, ref in ((R_in.grad, R64.grad), (T_in.grad, T64.grad), (v_in.grad, v64.grad)):
            assert float((got.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        # without gradient (rendering) the same numbers come out
        with torch.no_grad():
            R3, T3 = motion_basis(T(dst_Rs), T(dst_Ts), T(gt), T(rv))
        assert torch.equal(R3, Rs.detach()) and torch.equal(T3, Ts.detach())


def test_pose_mlp_kernels_match_fp64_autograd():
    """hnrf_pose_mlp_fwd / _bwd (BodyPoseRefiner.block_mlps on one pose vector, one workgroup each way) vs fp64
    autograd of the same MLP: the default 69-256x4-69, the freshly initialised refiner (last layer at 1e-5), odd shapes."""
    from humannerf_amd.network import BodyPoseRefiner, _PoseMLP
    for seed, dims in ((0, [69, 256, 256, 256, 256, 69]), (1, [5, 7, 3]), (2, [69, 128, 69]), (3, [12, 9]),
                       (4, [256] * 10)):
        rs = np.random.RandomState(seed)
        Ws = [(rs.randn(dims[l + 1], dims[l]) / np.sqrt(dims[l])).astype(np.float32) for l in range(len(dims) - 1)]
        bs = [(rs.randn(dims[l + 1]) * 0.1).astype(np.float32) for l in range(len(dims) - 1)]
        x = rs.randn(dims[0]).astype(np.float32)
        g = rs.randn(dims[-1]).astype(np.float32)
        T = lambda a: torch.from_numpy(a).to(dev())
        xg = T(x).requires_grad_(True)
        params = [p for W, b in zip(Ws, bs) for p in (T(W).requires_grad_(True), T(b).requires_grad_(True))]
        out = _PoseMLP.apply(xg, *params)
        (out * T(g)).sum().backward()
        x64 = torch.from_numpy(x).double().requires_grad_(True)
        p64 = [torch.from_numpy(a).double().requires_grad_(True) for W, b in zip(Ws, bs) for a in (W, b)]
        h = x64
        for l in range(len(Ws)):
            h = p64[2 * l] @ h + p64[2 * l + 1]
            if l + 1 < len(Ws):
                h = torch.relu(h)
        (h * torch.from_numpy(g).double()).sum().backward()
        assert float((out.detach().double().cpu() - h.detach()).abs().max()) <= 2e-6 * max(1.0, float(h.abs().max()))
        for got, ref in zip([xg] + params, [x64] + p64):
            scale = max(float(ref.grad.abs().max()), 1e-30)
            assert float((got.grad.double().cpu() - ref.grad).abs().max()) <= 5e-6 * scale, (dims, tuple(ref.shape))
    # the module route: fused rvec == the nn.Sequential it replaces, at the reference's initialisation
    torch.manual_seed(0)
    ref = BodyPoseRefiner().to(dev())
    pose = torch.randn(1, 69, device=dev()) * 0.3
    a = ref.rvec(pose)
    b = ref.block_mlps(pose).view(-1, 3)
    assert a.shape == (23, 3) and float((a - b).abs().max()) <= 1e-9 + 2e-6 * float(b.abs().max())
    a.square().sum().backward()
    ga = [p.grad.clone() for p in ref.parameters()]
    ref.zero_grad()
    b = ref.block_mlps(pose).view(-1, 3)
    b.square().sum().backward()
    for x1, x2 in zip(ga, [p.grad for p in ref.parameters()]):
        assert float((x1 - x2).abs().max()) <= 1e-5 * max(float(x2.abs().max()), 1e-30)


def test_decoder_layers_as_gemm_plus_fold_kernel():
    """network.conv_transpose3d_k4s2p1 on the GPU: one GEMM on the weight's native layout + hnrf_deconv_fold (the gather
    form of the stride-2 scatter) against torch's own F.conv_transpose3d evaluated in fp64 on the CPU, forward and all
    three gradients, for the decoder's layer shapes (network_util.py:30-42: 512->512 on 2^3, 512->256 on 4^3, 256->256
    on 8^3, 256->25 on 16^3) and a ragged one."""
    import torch.nn.functional as F
    from humannerf_amd.network import conv_transpose3d_k4s2p1
    torch.manual_seed(0)
    for cin, cout, dims in ((512, 512, (2, 2, 2)), (512, 256, (4, 4, 4)), (256, 256, (8, 8, 8)), (256, 25, (16, 16, 16)),
                            (7, 5, (3, 2, 4))):
        x = torch.randn(1, cin, *dims)
        w = torch.randn(cin, cout, 4, 4, 4) / np.sqrt(cin * 8.0)
        b = torch.randn(cout)
        g = torch.randn(1, cout, *[2 * d for d in dims])
        ref_in = [t.double().requires_grad_(True) for t in (x, w, b)]
        ref = F.conv_transpose3d(*ref_in, stride=2, padding=1)
        ref_g = torch.autograd.grad(ref, ref_in, g.double())
        got_in = [t.to(dev()).requires_grad_(True) for t in (x, w, b)]
        got = conv_transpose3d_k4s2p1(*got_in)
        got_g = torch.autograd.grad(got, got_in, g.to(dev()))
        assert got.shape == ref.shape
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        for a, r in zip(got_g, ref_g):
            assert float((a.cpu().double() - r).abs().max()) <= 2e-5 * float(r.abs().max()), (cin, cout, dims)
