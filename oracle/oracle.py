"""CPU ORACLE -- test infrastructure, NOT product code.

A plain restatement (torch CPU tensors, dtype-parametric fp32 / fp64) of the
reference's ray-marching hot path, SURVEY.md section 8(a) rows a1-a14.  Every function
cites the reference lines it follows.  It is pinned against outputs of the
reference itself (imported in the build container by ``oracle/make_golden.py``;
fixtures under ``tests/golden/``) by ``tests/test_oracle_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module -- as the checker / the reported CPU
baseline, never as part of the product path.  Nothing here touches a GPU and
nothing in ``humannerf_amd/`` imports it.

``state`` is a mapping with the reference's ``state_dict`` key names
("Checkpoint / resume").
"""
import math

import torch
import torch.nn.functional as F

SMPL_PARENT = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16,
               17, 18, 19, 20, 21]          # core/utils/network_util.py:91-94

DEFAULTS = dict(
    N_samples=128, perturb=0.0, chunk=32768, netchunk=300000, total_bones=24,
    nr_multires=6, nr_kick_in_iter=10000, nr_full_band_iter=50000,
    cnl_multires=10, pose_decoder_kick_in_iter=0, pose_decoder_off=False,
    ignore_non_rigid_motions=False, use_grid_sample=False,
)


def _lin(state, prefix, x):
    return F.linear(x, state[prefix + '.weight'].to(x.dtype), state[prefix + '.bias'].to(x.dtype))


# --------------------------------------------------------------------- a2
def rodrigues(rvec):
    """RodriguesModule.forward, core/utils/network_util.py:57-83."""
    theta = torch.sqrt(1e-5 + torch.sum(rvec ** 2, dim=1))
    r = rvec / theta[:, None]
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    return torch.stack((
        x * x + (1. - x * x) * c, x * y * (1. - c) - z * s, x * z * (1. - c) + y * s,
        x * y * (1. - c) + z * s, y * y + (1. - y * y) * c, y * z * (1. - c) - x * s,
        x * z * (1. - c) - y * s, y * z * (1. - c) + x * s, z * z + (1. - z * z) * c),
        dim=1).view(-1, 3, 3)


def pose_refine(state, dst_posevec, dst_Rs):
    """BodyPoseRefiner (pose_decoders/mlp_delta_body_pose.py:35-41) and its
    application network.py:672-688: dst_Rs[1:] <- dst_Rs[1:] @ dR."""
    h = dst_posevec[None]
    n = len([k for k in state if k.startswith('pose_decoder.block_mlps.') and k.endswith('.weight')])
    for i in range(n):
        h = _lin(state, f'pose_decoder.block_mlps.{2 * i}', h)
        if i < n - 1:
            h = torch.relu(h)
    dR = rodrigues(h.view(-1, 3))                      # (23,3,3)
    out = dst_Rs.clone()
    out[1:] = torch.matmul(dst_Rs[1:], dR)
    return out


# --------------------------------------------------------------------- a3
def motion_basis(dst_Rs, dst_Ts, cnl_gtfms):
    """MotionBasisComputer.forward, core/utils/network_util.py:125-156."""
    B = dst_Rs.shape[0]
    G = torch.zeros(B, 4, 4, dtype=dst_Rs.dtype, device=dst_Rs.device)
    G[:, :3, :3] = dst_Rs
    G[:, :3, 3] = dst_Ts
    G[:, 3, 3] = 1.0
    A = [G[0]]
    for i in range(1, B):
        A.append(torch.matmul(A[SMPL_PARENT[i]], G[i]))
    A = torch.stack(A)
    Fm = torch.matmul(cnl_gtfms, torch.inverse(A))
    return Fm[:, :3, :3], Fm[:, :3, 3]


# --------------------------------------------------------------------- a4
def weight_volume(state, priors):
    """MotionWeightVolumeDecoder.forward (mweight_vol_decoders/
    deconv_vol_decoder.py:25-33) over ConvDecoder3D (network_util.py:12-50)."""
    dt = priors.dtype
    p = 'mweight_vol_decoder.'
    h = F.leaky_relu(_lin(state, p + 'decoder.block_mlp.0', state[p + 'const_embedding'].to(dt)[None]), 0.2)
    h = h.view(-1, 1024, 1, 1, 1)
    idx = sorted(int(k.split('.')[3]) for k in state
                 if k.startswith(p + 'decoder.block_conv.') and k.endswith('.weight'))
    for n, i in enumerate(idx):
        h = F.conv_transpose3d(h, state[p + f'decoder.block_conv.{i}.weight'].to(dt),
                               state[p + f'decoder.block_conv.{i}.bias'].to(dt), stride=2, padding=1)
        if n < len(idx) - 1:
            h = F.leaky_relu(h, 0.2)
    return F.softmax(h + torch.log(priors[None]), dim=1)[0]


# --------------------------------------------------------------------- a6
def linspace01(S, dtype, device=None):
    """torch.linspace(0., 1., steps=S) as used at network.py:457."""
    return torch.linspace(0., 1., steps=S, dtype=dtype, device=device)


def z_values(near, far, S, t_rand=None):
    """_get_samples_along_ray + _stratified_sampling, network.py:455-471.
    near/far: (R,1).  ``t_rand`` (R,S) in [0,1) replaces torch.rand."""
    t = linspace01(S, near.dtype, near.device)
    z = near * (1. - t) + far * t
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


# --------------------------------------------------------------------- a8
def trilinear_zeros(vol, g):
    """F.grid_sample(vol[None,None], g, padding_mode='zeros',
    align_corners=True) for one (D,H,W) volume and (P,3) grid points in
    [-1,1], g = (x,y,z) <-> vol[z][y][x] (network.py:409-413).  Written out
    corner by corner; out-of-range corners contribute zero."""
    D, H, W = vol.shape
    ix = (g[:, 0] + 1.) * 0.5 * (W - 1)
    iy = (g[:, 1] + 1.) * 0.5 * (H - 1)
    iz = (g[:, 2] + 1.) * 0.5 * (D - 1)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    fx, fy, fz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = torch.zeros_like(ix)
    flat = vol.reshape(-1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                w = (fx if dx else 1. - fx) * (fy if dy else 1. - fy) * (fz if dz else 1. - fz)
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                lin = (zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1)
                out = out + torch.where(ok, flat[lin] * w, torch.zeros_like(w))
    return out


def sample_motion_fields(pts, Rs, Ts, vol, bbox_min, bbox_scale, use_grid_sample=False):
    """_sample_motion_fields, network.py:392-444.  pts (P,3); vol (B+1,D,H,W)
    with the background channel last (dropped, 404).  Returns x_skel (P,3),
    fg mask = sum w (P,), unnormalised weights (P,B).  ``use_grid_sample``: call
    F.grid_sample like the reference does (the eager-GPU timing comparator) instead
    of the written-out trilinear (the independent checker)."""
    nb = vol.shape[0] - 1
    ws, poss = [], []
    for i in range(nb):
        pos = torch.matmul(Rs[i], pts.T).T + Ts[i]
        g = (pos - bbox_min[None]) * bbox_scale[None] - 1.0
        if use_grid_sample:
            ws.append(F.grid_sample(vol[None, i:i + 1], g[None, None, None], padding_mode='zeros',
                                    align_corners=True)[0, 0, 0, 0])
        else:
            ws.append(trilinear_zeros(vol[i], g))
        poss.append(pos)
    w = torch.stack(ws, dim=-1)                        # (P,B)
    wsum = torch.sum(w, dim=-1, keepdim=True)
    x_skel = torch.sum(torch.stack([w[:, i:i + 1] * poss[i] for i in range(nb)], 0), 0) \
        / wsum.clamp(min=0.0001)
    return x_skel, wsum[:, 0], w


# --------------------------------------------------------------- a10 / a12
def hann_weights(iter_val, multires, kick_in_iter, full_band_iter, dtype=torch.float32):
    """Window weights of embedders/hannw_fourier.py:26-40 (computed in fp32
    like the reference: the constants there are float32 tensors)."""
    kick = torch.tensor(float(kick_in_iter), dtype=torch.float32)
    t = torch.clamp(torch.as_tensor(iter_val, dtype=torch.float32).reshape(()) - kick, min=0.)
    N = full_band_iter - kick_in_iter
    alpha = torch.tensor(float(multires), dtype=torch.float32) if N == 0 else multires * t / N
    k = torch.arange(multires, dtype=torch.float32)
    w = (1. - torch.cos(math.pi * torch.clamp(alpha - k, min=0., max=1.))) / 2.
    return w.to(dtype)


def hann_pe(x, hann_w):
    """hannw_fourier embed: [w_k sin(2^k x), w_k cos(2^k x)]_k, no input term
    (hannw_fourier.py:21-49)."""
    out = []
    for k in range(hann_w.shape[0]):
        f = 2.0 ** k
        out += [hann_w[k] * torch.sin(x * f), hann_w[k] * torch.cos(x * f)]
    return torch.cat(out, -1)


def fourier_pe(x, multires):
    """fourier embed with include_input (embedders/fourier.py:9-38)."""
    out = [x]
    for k in range(multires):
        f = 2.0 ** k
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


# -------------------------------------------------------------- a11 / a13
def non_rigid_mlp(state, pe, cond, x_skel):
    """NonRigidMotionMLP.forward, non_rigid_motion_mlps/mlp_offset.py:74-114
    (default branch).  Skip input order is [h, pe]."""
    p = 'non_rigid_mlp.module.block_mlps.'
    idx = sorted(int(k[len(p):].split('.')[0]) for k in state if k.startswith(p) and k.endswith('.weight'))
    h = torch.cat([cond.expand(pe.shape[0], -1), pe], dim=-1)
    for n, i in enumerate(idx):
        W = state[p + f'{i}.weight']
        if n > 0 and W.shape[1] == h.shape[1] + pe.shape[1]:
            h = torch.cat([h, pe], dim=-1)             # layers_to_cat_inputs
        h = _lin(state, p + str(i), h)
        if n < len(idx) - 1:
            h = torch.relu(h)
    return x_skel + h, h


def canonical_mlp(state, pe):
    """CanonicalMLP.forward default branch, canonical_mlps/mlp_rgb_sigma.py:
    132-198.  Skip input order is [pe, h]."""
    p = 'cnl_mlp.module.pts_linears.'
    idx = sorted(int(k[len(p):].split('.')[0]) for k in state if k.startswith(p) and k.endswith('.weight'))
    h = pe
    for n, i in enumerate(idx):
        W = state[p + f'{i}.weight']
        if n > 0 and W.shape[1] == h.shape[1] + pe.shape[1]:
            h = torch.cat([pe, h], dim=-1)             # layers_to_cat_input
        h = torch.relu(_lin(state, p + str(i), h))
    return _lin(state, 'cnl_mlp.module.output_linear.0', h)


# -------------------------------------------------------------------- a14
def raw2outputs(raw, mask, z, rays_d, xyz, bgcolor):
    """_raw2outputs, network.py:355-388.  raw (R,S,4), mask (R,S), z (R,S)."""
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = (1.0 - torch.exp(-torch.relu(raw[..., 3]) * dists)) * mask
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * T
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth = torch.sum(weights * z, -1)
    acc = torch.sum(weights, -1)
    rgb_map = rgb_map + (1. - acc[..., None]) * bgcolor[None, :] / 255.
    wmax, ind = weights.max(dim=1)
    ind3 = ind[:, None, None].expand(-1, 1, 3)
    return dict(rgb=rgb_map, alpha=acc, depth=depth, weights_on_rays=weights,
                rgb_on_rays=rgb, cnl_xyz=torch.gather(xyz, 1, ind3).squeeze(1),
                cnl_rgb=torch.gather(rgb, 1, ind3).squeeze(1), cnl_weight=wmax)


# --------------------------------------------------------------------- a1
def per_frame_setup(state, data, iter_val, opt):
    """The per-frame part of Network.forward, network.py:659-763."""
    dt = data['rays'].dtype
    dst_Rs, dst_Ts = data['dst_Rs'].to(dt), data['dst_Ts'].to(dt)
    posevec = data['dst_posevec'].to(dt)
    if iter_val >= opt['pose_decoder_kick_in_iter'] and not opt['pose_decoder_off']:
        dst_Rs = pose_refine(state, posevec, dst_Rs)
    hw = hann_weights(iter_val, opt['nr_multires'], opt['nr_kick_in_iter'], opt['nr_full_band_iter'], dt).to(dst_Rs.device)
    cond = posevec[None]
    if iter_val < opt['nr_kick_in_iter']:
        cond = torch.zeros_like(cond) * cond           # network.py:735-737
    Rs, Ts = motion_basis(dst_Rs, dst_Ts, data['cnl_gtfms'].to(dt))
    vol = weight_volume(state, data['motion_weights_priors'].to(dt))
    return dict(Rs=Rs, Ts=Ts, vol=vol, hann_w=hw, cond=cond)


def render_rays(state, fr, rays_o, rays_d, near, far, bbox_min, bbox_scale, bgcolor, opt, t_rand=None):
    """_render_rays for one ray chunk, network.py:474-602 (default branches)."""
    S = opt['N_samples']
    R = rays_o.shape[0]
    z = z_values(near, far, S, t_rand)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    x_skel, mask, bmw = sample_motion_fields(pts.reshape(-1, 3), fr['Rs'], fr['Ts'], fr['vol'], bbox_min, bbox_scale,
                                             opt['use_grid_sample'])
    raws, xyzs, offs = [], [], []
    for s in range(0, x_skel.shape[0], opt['netchunk']):        # network.py:252
        xs = x_skel[s:s + opt['netchunk']]
        if not opt['ignore_non_rigid_motions']:
            xyz, ofs = non_rigid_mlp(state, hann_pe(xs, fr['hann_w']), fr['cond'], xs)
        else:
            xyz, ofs = xs, torch.zeros_like(xs)
        raws.append(canonical_mlp(state, fourier_pe(xyz, opt['cnl_multires'])))
        xyzs.append(xyz)
        offs.append(ofs)
    raw = torch.cat(raws).view(R, S, 4)
    xyz = torch.cat(xyzs).view(R, S, 3)
    out = raw2outputs(raw, mask.view(R, S), z, rays_d, xyz, bgcolor)
    out.update(xyz_on_rays=xyz, backward_motion_weights=bmw.view(R, S, -1),
               offsets=torch.cat(offs).view(R, S, 3))
    # extra stage intermediates (not part of the reference's output dict)
    out.update(_x_skel=x_skel.view(R, S, 3), _mask=mask.view(R, S), _raw=raw, _z_vals=z)
    return out


def render(state, data, iter_val=1e7, t_rand=None, dtype=torch.float32, device=None, **overrides):
    """Network.forward, network.py:647-789.  ``data`` holds the per-frame
    tensors of row a1 (numpy arrays or tensors).  Returns the 11 reference
    keys (+ underscore-prefixed intermediates).  ``device``: where the eager
    torch ops run (default CPU; a GPU device gives the "reference op sequence in
    PyTorch-ROCm eager" comparator of SURVEY.md section 8d)."""
    opt = dict(DEFAULTS)
    opt.update(overrides)
    d = {k: (torch.as_tensor(v).to(dtype) if torch.as_tensor(v).is_floating_point() else torch.as_tensor(v))
         for k, v in data.items() if k in ('rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms',
                                          'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz',
                                          'cnl_bbox_scale_xyz', 'bgcolor')}
    state = {k: torch.as_tensor(v).to(dtype) for k, v in state.items()}
    if device is not None:
        d = {k: v.to(device) for k, v in d.items()}
        state = {k: v.to(device) for k, v in state.items()}
        if t_rand is not None:
            t_rand = torch.as_tensor(t_rand).to(device)
    fr = per_frame_setup(state, d, iter_val, opt)
    rays_o, rays_d = d['rays'][0].reshape(-1, 3), d['rays'][1].reshape(-1, 3)
    outs = []
    for s in range(0, rays_o.shape[0], opt['chunk']):          # network.py:333
        e = s + opt['chunk']
        outs.append(render_rays(state, fr, rays_o[s:e], rays_d[s:e], d['near'][s:e], d['far'][s:e],
                                d['cnl_bbox_min_xyz'], d['cnl_bbox_scale_xyz'], d['bgcolor'], opt,
                                None if t_rand is None else torch.as_tensor(t_rand).to(dtype)[s:e]))
    out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    out['_vol'] = fr['vol']
    out['_Rs'], out['_Ts'] = fr['Rs'], fr['Ts']
    return out
