"""Host-side operators of the hot path: thin torch-tensor wrappers over the C ABI.

PyTorch is plumbing here (device memory, streams); all arithmetic of the path
happens in libhnrf.so.  Every op requires CUDA(ROCm) fp32 contiguous tensors and
raises otherwise -- there is no eager fallback.
"""
import ctypes

import torch

from . import _lib

MLP_MODES = {'f32': 0, 'f16x3': 1}
# training entry points only: split-f16 arithmetic with the saved weight-gradient operands (activations, dZ) in f16
TRAIN_MODES = {'f32': 0, 'f16x3': 1, 'f16x3h': 2}


NO_RANGE_GUARD, GUARD_ONE_CHUNK = 0x100, 0x200          # hnrf.h: HNRF_MLP_NO_RANGE_GUARD / HNRF_MLP_GUARD_ONE_CHUNK


def _mode_arg(mode):
    """``mode`` argument of the forward entry points: 'f32' | 'f16x3', optionally with the f16-range guard selection
    (hnrf.h) appended -- 'f16x3+noguard' (unguarded kernel instances) or 'f16x3+guard1:<k>' (hnrf_render_frame_fwd:
    only ray chunk k % n_chunks is guarded).  Plain names guard every launch."""
    base, _, opt = mode.partition('+')
    m = MLP_MODES[base]
    if not opt:
        return m
    if opt == 'noguard':
        return m | NO_RANGE_GUARD
    if opt.startswith('guard1:'):
        return m | GUARD_ONE_CHUNK | ((int(opt[7:]) & 0x7fff) << 16)
    raise ValueError('unknown mode option %r' % mode)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.HnrfError(
                f'hot-path operand must be a contiguous fp32 tensor on the GPU, got '
                f'{t.dtype} {t.device} contiguous={t.is_contiguous()} shape={tuple(t.shape)}')


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def sample_warp(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                n_samples, want_bmw=False, bmw_out=None):
    """K1 (network.py:455-471, 499, 392-444).  Returns z_vals (R,S), x_skel (R,S,3),
    fg_mask (R,S), bmw (R,S,B) or None.  ``bmw_out``: contiguous (R,S,B) tensor to write bmw into."""
    lib = _lib.load()
    near, far = near.reshape(-1), far.reshape(-1)
    _chk(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale)
    R, S = rays_o.shape[0], int(n_samples)
    B = motion_Rs.shape[0]
    G = vol.shape[-1]
    assert vol.shape[0] >= B and vol.shape[1] == vol.shape[2] == G
    assert rays_d.shape == (R, 3) and near.numel() == R and far.numel() == R
    if t_rand is not None:
        assert t_rand.shape == (R, S)
    dev = rays_o.device
    z = torch.empty(R, S, device=dev)
    x_skel = torch.empty(R, S, 3, device=dev)
    mask = torch.empty(R, S, device=dev)
    bmw = (bmw_out if bmw_out is not None else torch.empty(R, S, B, device=dev)) if want_bmw else None
    if bmw is not None:
        assert bmw.shape == (R, S, B)
        _chk(bmw)
    _lib.check(lib.hnrf_sample_warp_fwd(_ptr(rays_o), _ptr(rays_d), _ptr(near), _ptr(far), _ptr(t_rand),
                                        _ptr(motion_Rs), _ptr(motion_Ts), _ptr(vol), _ptr(bbox_min),
                                        _ptr(bbox_scale), R, S, B, G, _ptr(z), _ptr(x_skel), _ptr(mask),
                                        _ptr(bmw), _stream()), 'hnrf_sample_warp_fwd')
    return z, x_skel, mask, bmw


def nonrigid_pack(weights, biases, cond, mode='f32', out=None):
    """Pack block_mlps.{0..12} (7 Linear layers) + the frame's condition code."""
    lib = _lib.load()
    m = MLP_MODES[mode]
    cond = cond.reshape(-1)
    _chk(*weights, *biases, cond)
    assert len(weights) == 7 and len(biases) == 7 and cond.numel() == 69
    shapes = [(128, 105), (128, 128), (128, 128), (128, 128), (128, 164), (128, 128), (3, 128)]
    for w, b, s in zip(weights, biases, shapes):
        if tuple(w.shape) != s or b.numel() != s[0]:
            raise _lib.HnrfError(f'non-rigid MLP layer shape {tuple(w.shape)} != {s}: only the default '
                                 f'architecture (default.yaml:142-165) is built')
    nbytes = lib.hnrf_nonrigid_packed_bytes(m)
    if out is None or out.numel() * 4 < nbytes:
        out = torch.empty((nbytes + 3) // 4, device=cond.device)
    _lib.check(lib.hnrf_nonrigid_pack(_ptr_array(weights), _ptr_array(biases), _ptr(cond), m, _ptr(out),
                                      _stream()), 'hnrf_nonrigid_pack')
    return out


def nonrigid(x_skel, hann_w, packed, mode='f32', want_offsets=False, xyz_out=None, offsets_out=None):
    """K2 (hannw_fourier.py:21-49 + mlp_offset.py:74-114).  x_skel (...,3).  ``xyz_out`` / ``offsets_out``:
    contiguous tensors of x_skel's shape to write into."""
    lib = _lib.load()
    _chk(x_skel, hann_w, packed, xyz_out, offsets_out)
    P = x_skel.numel() // 3
    xyz = xyz_out if xyz_out is not None else torch.empty_like(x_skel)
    offsets = (offsets_out if offsets_out is not None else torch.empty_like(x_skel)) if want_offsets else None
    assert xyz.shape == x_skel.shape and (offsets is None or offsets.shape == x_skel.shape)
    _lib.check(lib.hnrf_nonrigid_fwd(_ptr(x_skel), _ptr(hann_w), _ptr(packed), _mode_arg(mode), P, _ptr(xyz),
                                     _ptr(offsets), _stream()), 'hnrf_nonrigid_fwd')
    return xyz, offsets


def canonical_pack(weights, biases, mode='f32', out=None):
    """Pack pts_linears.{0..14} (8 layers) + output_linear.0."""
    lib = _lib.load()
    m = MLP_MODES[mode]
    _chk(*weights, *biases)
    assert len(weights) == 9 and len(biases) == 9
    shapes = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 2 + [(4, 256)]
    for w, b, s in zip(weights, biases, shapes):
        if tuple(w.shape) != s or b.numel() != s[0]:
            raise _lib.HnrfError(f'canonical MLP layer shape {tuple(w.shape)} != {s}: only the default '
                                 f'architecture (default.yaml:51-57) is built')
    nbytes = lib.hnrf_canonical_packed_bytes(m)
    if out is None or out.numel() * 4 < nbytes:
        out = torch.empty((nbytes + 3) // 4, device=weights[0].device)
    _lib.check(lib.hnrf_canonical_pack(_ptr_array(weights), _ptr_array(biases), m, _ptr(out), _stream()),
               'hnrf_canonical_pack')
    return out


def status_word(packed, which, mode):
    """View (1 int32 element, on the device) of a packed image's status word, or None when ``mode`` has none
    (hnrf_canonical_status_offset / hnrf_nonrigid_status_offset): the f16x3 inference kernels OR
    STATUS_F16_RANGE into it when an activation came within reach of the f16 clamp."""
    lib = _lib.load()
    off = (lib.hnrf_canonical_status_offset if which == 'canonical' else lib.hnrf_nonrigid_status_offset)(MLP_MODES[mode.partition('+')[0]])
    if off == 0:
        return None
    assert off % 4 == 0 and packed.dtype == torch.float32
    return packed.view(torch.int32)[off // 4:off // 4 + 1]


STATUS_F16_RANGE = 1


def canonical(xyz, packed, mode='f32'):
    """K3 (fourier.py:9-38 + mlp_rgb_sigma.py:132-198).  xyz (...,3) -> raw (...,4)."""
    lib = _lib.load()
    _chk(xyz, packed)
    P = xyz.numel() // 3
    raw = torch.empty(*xyz.shape[:-1], 4, device=xyz.device)
    _lib.check(lib.hnrf_canonical_fwd(_ptr(xyz), _ptr(packed), _mode_arg(mode), P, _ptr(raw), _stream()),
               'hnrf_canonical_fwd')
    return raw


def composite(raw, fg_mask, z_vals, rays_d, xyz, bgcolor, diagnostics=True, cull_eps=0.0, out=None):
    """K4 (network.py:355-388).  Returns dict with rgb/alpha/depth and, when
    ``diagnostics``, weights_on_rays, rgb_on_rays, cnl_xyz, cnl_rgb, cnl_weight.  ``out``: dict of contiguous
    tensors of those shapes to write into (e.g. row ranges of whole-frame buffers)."""
    lib = _lib.load()
    _chk(raw, fg_mask, z_vals, rays_d, xyz, bgcolor)
    R, S = z_vals.shape
    dev = raw.device
    shapes = {'rgb': (R, 3), 'alpha': (R,), 'depth': (R,)}
    if diagnostics:
        shapes.update(weights_on_rays=(R, S), rgb_on_rays=(R, S, 3), cnl_xyz=(R, 3), cnl_rgb=(R, 3), cnl_weight=(R,))
    if out is None:
        out = {k: torch.empty(*sh, device=dev) for k, sh in shapes.items()}
    else:
        out = {k: out[k] for k in shapes}
        for k, sh in shapes.items():
            assert tuple(out[k].shape) == sh, (k, tuple(out[k].shape), sh)
        _chk(*out.values())
    g = out.get
    _lib.check(lib.hnrf_composite_fwd(_ptr(raw), _ptr(fg_mask), _ptr(z_vals), _ptr(rays_d), _ptr(xyz),
                                      _ptr(bgcolor), R, S, float(cull_eps), _ptr(out['rgb']), _ptr(out['alpha']),
                                      _ptr(out['depth']), _ptr(g('weights_on_rays')), _ptr(g('rgb_on_rays')),
                                      _ptr(g('cnl_xyz')), _ptr(g('cnl_rgb')), _ptr(g('cnl_weight')), _stream()),
               'hnrf_composite_fwd')
    return out


def render_workspace_bytes(R, S):
    return _lib.load().hnrf_render_workspace_bytes(int(R), int(S))


def render_term_workspace_bytes(R, S):
    return _lib.load().hnrf_render_term_workspace_bytes(int(R), int(S))


def render_rays(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                hann_w, nr_packed, cnl_packed, bgcolor, n_samples, mode='f32', workspace=None, out=None,
                mlp_events=None, cull_eps=0.0):
    """The whole path for one ray chunk (network.py:474-602) with only the
    rgb/alpha/depth outputs; intermediates live in ``workspace``.  ``mlp_events``:
    optional pair of torch.cuda.Event(enable_timing=True), recorded around the
    canonical-MLP launch."""
    lib = _lib.load()
    near, far = near.reshape(-1), far.reshape(-1)
    _chk(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w,
         nr_packed, cnl_packed, bgcolor)
    R, S = rays_o.shape[0], int(n_samples)
    need = lib.hnrf_render_workspace_bytes(R, S)
    if workspace is None:
        workspace = torch.empty(need // 4 + 64, device=rays_o.device)
    assert workspace.numel() * workspace.element_size() >= need
    dev = rays_o.device
    if out is None:
        out = {'rgb': torch.empty(R, 3, device=dev), 'alpha': torch.empty(R, device=dev),
               'depth': torch.empty(R, device=dev)}
    ev = (0, 0)
    if mlp_events is not None:
        for e in mlp_events:
            if not e.cuda_event:
                e.record()            # forces creation of the hipEvent_t
        ev = (mlp_events[0].cuda_event, mlp_events[1].cuda_event)
    _lib.check(lib.hnrf_render_rays_fwd(_ptr(rays_o), _ptr(rays_d), _ptr(near), _ptr(far), _ptr(t_rand),
                                        _ptr(motion_Rs), _ptr(motion_Ts), _ptr(vol), _ptr(bbox_min),
                                        _ptr(bbox_scale), _ptr(hann_w), _ptr(nr_packed), _ptr(cnl_packed),
                                        _ptr(bgcolor), _mode_arg(mode), float(cull_eps), R, S, motion_Rs.shape[0], vol.shape[-1],
                                        _ptr(workspace), workspace.numel() * workspace.element_size(),
                                        _ptr(out['rgb']), _ptr(out['alpha']), _ptr(out['depth']),
                                        ev[0], ev[1], _stream()),
               'hnrf_render_rays_fwd')
    return out


_frame_ctx = {}


def _frame_streams(device):
    """Per-device side stream + the 5 events hnrf_render_frame_fwd orders the two streams with."""
    ctx = _frame_ctx.get(device.index)
    if ctx is None:
        side = torch.cuda.Stream(device=device)
        evs = [torch.cuda.Event() for _ in range(5)]
        for e in evs:
            e.record()                       # forces creation of the hipEvent_t
        ctx = _frame_ctx[device.index] = (side, evs)
    return ctx


def render_frame(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w, nr_packed,
                 cnl_packed, bgcolor, n_samples, chunk, mode='f16x3', diagnostics=True, cull_eps=0.0, workspace=None,
                 overlap=True, mlp_event_log=None):
    """The whole frame in one call (hnrf_render_frame_fwd): all ray chunks through K1..K4, results in whole-frame tensors;
    K1 of the next chunk on a side stream while the MLP kernels of the current one run.  Returns (dict of outputs,
    workspace).  ``mlp_event_log``: list that receives one (start, stop) torch.cuda.Event pair per chunk."""
    lib = _lib.load()
    near, far = near.reshape(-1), far.reshape(-1)
    _chk(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w, nr_packed, cnl_packed,
         bgcolor)
    N, S, B, G = rays_o.shape[0], int(n_samples), motion_Rs.shape[0], vol.shape[-1]
    dev = rays_o.device
    chunk = int(chunk)
    need = lib.hnrf_render_frame_workspace_bytes(min(chunk, max(N, 1)), S)
    if workspace is None or workspace.numel() * 4 < need or workspace.device != dev:
        workspace = torch.empty(need // 4 + 64, device=dev)
    shp = {'rgb': (3,), 'alpha': (), 'depth': ()}
    if diagnostics:
        shp.update(weights_on_rays=(S,), rgb_on_rays=(S, 3), cnl_xyz=(3,), cnl_rgb=(3,), cnl_weight=(), xyz_on_rays=(S, 3),
                   backward_motion_weights=(S, B), offsets=(S, 3))
    out = {k: torch.empty((N,) + v, device=dev) for k, v in shp.items()}
    g = lambda k: _ptr(out.get(k))
    side, evs, ev_arr = None, None, None
    if overlap and N > chunk:
        side, evs = _frame_streams(dev)
        ev_arr = (ctypes.c_void_p * 5)(*[e.cuda_event for e in evs])
    mlp_arr, pairs = None, []
    if mlp_event_log is not None:
        nchunk = (N + chunk - 1) // chunk
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nchunk)]
        for a, b in pairs:
            a.record()
            b.record()
        mlp_arr = (ctypes.c_void_p * (2 * nchunk))(*[e.cuda_event for p in pairs for e in p])
    _lib.check(lib.hnrf_render_frame_fwd(
        _ptr(rays_o), _ptr(rays_d), _ptr(near), _ptr(far), _ptr(t_rand), _ptr(motion_Rs), _ptr(motion_Ts), _ptr(vol),
        _ptr(bbox_min), _ptr(bbox_scale), _ptr(hann_w), _ptr(nr_packed), _ptr(cnl_packed), _ptr(bgcolor), _mode_arg(mode),
        float(cull_eps), N, S, B, G, chunk, _ptr(workspace), workspace.numel() * 4, g('rgb'), g('alpha'), g('depth'),
        g('weights_on_rays'), g('rgb_on_rays'), g('cnl_xyz'), g('cnl_rgb'), g('cnl_weight'), g('xyz_on_rays'),
        g('backward_motion_weights'), g('offsets'), side.cuda_stream if side is not None else None, ev_arr, mlp_arr,
        _stream()), 'hnrf_render_frame_fwd')
    if side is not None:
        # the side stream read the inputs and wrote backward_motion_weights / the workspace: keep the allocator from
        # handing those blocks out again before it is done (everything it did is also ordered on the main stream)
        for t in (rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, workspace,
                  out.get('backward_motion_weights')):
            if t is not None:
                t.record_stream(side)
    if mlp_event_log is not None:
        mlp_event_log.extend(pairs)
    return out, workspace


def render_rays_term(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                     hann_w, nr_packed, cnl_packed, bgcolor, n_samples, mode='f16x3', term_eps=1e-4, cull_eps=0.0,
                     workspace=None, want_count=False):
    """Lean path with early ray termination (opt-in approximation, |d rgb|, |d alpha| <= term_eps): see
    hnrf_render_rays_term_fwd.  Returns the rgb/alpha/depth dict (+ 'evaluated': device int tensor when asked)."""
    lib = _lib.load()
    near, far = near.reshape(-1), far.reshape(-1)
    _chk(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w,
         nr_packed, cnl_packed, bgcolor)
    R, S = rays_o.shape[0], int(n_samples)
    need = lib.hnrf_render_term_workspace_bytes(R, S)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 4 + 64, device=rays_o.device)
    dev = rays_o.device
    out = {'rgb': torch.empty(R, 3, device=dev), 'alpha': torch.empty(R, device=dev), 'depth': torch.empty(R, device=dev)}
    ev = torch.empty(1, dtype=torch.int32, device=dev) if want_count else None
    _lib.check(lib.hnrf_render_rays_term_fwd(_ptr(rays_o), _ptr(rays_d), _ptr(near), _ptr(far), _ptr(t_rand),
                                             _ptr(motion_Rs), _ptr(motion_Ts), _ptr(vol), _ptr(bbox_min),
                                             _ptr(bbox_scale), _ptr(hann_w), _ptr(nr_packed), _ptr(cnl_packed),
                                             _ptr(bgcolor), _mode_arg(mode), float(cull_eps), float(term_eps), R, S,
                                             motion_Rs.shape[0], vol.shape[-1], _ptr(workspace),
                                             workspace.numel() * workspace.element_size(), _ptr(out['rgb']),
                                             _ptr(out['alpha']), _ptr(out['depth']),
                                             None if ev is None else ev.data_ptr(), _stream()),
               'hnrf_render_rays_term_fwd')
    if ev is not None:
        out['evaluated'] = ev
    return out


# ----------------------------------------------------------------------------- training
def canonical_train(xyz, packed, mode='f32'):
    """hnrf_canonical_fwd_train: raw (...,4), pe (P,63), acts (8,P,256), relu sign masks (8,P,8) int32.
    ``packed`` must have been made for the same arithmetic (the 'f16x3' image serves 'f16x3h').  mode 'f16x3h':
    acts is float16 in the BLOCKED layout of hnrf_mlp_dw_h, (8, P rounded up to 128, 256) -- opaque, for mlp_dw_h(...,
    x_blocked=True) -- and pe a row-major float16 (P,64) matrix whose last column is zero."""
    lib = _lib.load()
    _chk(xyz, packed)
    P = xyz.numel() // 3
    dev = xyz.device
    half = mode == 'f16x3h'
    raw = torch.empty(*xyz.shape[:-1], 4, device=dev)
    pe = torch.empty(P, 64, device=dev, dtype=torch.float16) if half else torch.empty(P, 63, device=dev)
    acts = torch.empty(8, (P + 127) // 128 * 128, 256, device=dev, dtype=torch.float16) if half else torch.empty(8, P, 256, device=dev)
    bits = torch.empty(8, P, 8, dtype=torch.int32, device=dev)
    _lib.check(lib.hnrf_canonical_fwd_train(_ptr(xyz), _ptr(packed), TRAIN_MODES[mode], P, _ptr(raw), _ptr(pe),
                                            _ptr(acts), bits.data_ptr(), _stream()), 'hnrf_canonical_fwd_train')
    return raw, pe, acts, bits


def nonrigid_train(x_skel, hann_w, packed, mode='f32'):
    """hnrf_nonrigid_fwd_train: xyz, offsets, pe (P,36), acts (6,P,128), relu sign masks (6,P,4) int32; mode 'f16x3h':
    float16 acts and a float16 pe (P,64) with columns 36.. zero."""
    lib = _lib.load()
    _chk(x_skel, hann_w, packed)
    P = x_skel.numel() // 3
    dev = x_skel.device
    half = mode == 'f16x3h'
    xyz, offsets = torch.empty_like(x_skel), torch.empty_like(x_skel)
    pe = torch.empty(P, 64, device=dev, dtype=torch.float16) if half else torch.empty(P, 36, device=dev)
    acts = torch.empty(6, (P + 127) // 128 * 128, 128, device=dev, dtype=torch.float16) if half else torch.empty(6, P, 128, device=dev)
    bits = torch.empty(6, P, 4, dtype=torch.int32, device=dev)
    _lib.check(lib.hnrf_nonrigid_fwd_train(_ptr(x_skel), _ptr(hann_w), _ptr(packed), TRAIN_MODES[mode], P, _ptr(xyz),
                                           _ptr(offsets), _ptr(pe), _ptr(acts), bits.data_ptr(), _stream()),
               'hnrf_nonrigid_fwd_train')
    return xyz, offsets, pe, acts, bits


def composite_bwd(raw, fg_mask, z_vals, rays_d, bgcolor, g_rgb, g_alpha=None, g_depth=None):
    lib = _lib.load()
    _chk(raw, fg_mask, z_vals, rays_d, bgcolor, g_rgb, g_alpha, g_depth)
    R, S = z_vals.shape
    d_raw, d_mask = torch.empty_like(raw), torch.empty_like(fg_mask)
    _lib.check(lib.hnrf_composite_bwd(_ptr(raw), _ptr(fg_mask), _ptr(z_vals), _ptr(rays_d), _ptr(bgcolor), _ptr(g_rgb),
                                      _ptr(g_alpha), _ptr(g_depth), R, S, _ptr(d_raw), _ptr(d_mask), _stream()),
               'hnrf_composite_bwd')
    return d_raw, d_mask


def pe_bwd(x, g, hann_w, n_bands, include_input, out=None):
    """d PE -> d position; accumulates into ``out`` when given."""
    lib = _lib.load()
    _chk(x, g, hann_w, out)
    P = x.numel() // 3
    assert g.shape[-1] == (3 if include_input else 0) + 6 * n_bands and g.numel() // g.shape[-1] == P
    acc = out is not None
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.hnrf_pe_bwd(_ptr(x), _ptr(g), _ptr(hann_w), P, n_bands, int(include_input), int(acc), _ptr(out),
                               _stream()), 'hnrf_pe_bwd')
    return out


def canonical_bwd(xyz, d_raw, bits, weights, mode='f32'):
    """dX chain of the canonical MLP: returns dZ (8,P,256), d_xyz (P,3) and amax (8,64) (max over row l bounds
    |dZ_l|).  bits: sign masks from canonical_train.  mode 'f16x3h': dZ is float16 in the chain's scaled domain and the
    third result is scale (8,): the power of two dZ_l was multiplied by (mlp_dw_h divides it out)."""
    lib = _lib.load()
    _chk(xyz, d_raw, *weights)
    P = xyz.numel() // 3
    assert bits.shape == (8, P, 8) and bits.dtype == torch.int32 and bits.is_contiguous()
    assert d_raw.numel() == 4 * P and len(weights) == 9
    half = mode == 'f16x3h'
    m = MLP_MODES['f16x3' if half else mode]
    packed = torch.empty((lib.hnrf_canonical_bwd_packed_bytes(m) + 3) // 4, device=xyz.device)
    _lib.check(lib.hnrf_canonical_bwd_pack(_ptr_array(weights), m, _ptr(packed), _stream()), 'hnrf_canonical_bwd_pack')
    d_raw_amax = d_raw.abs().amax().reshape(1) if mode != 'f32' else None
    dZ = torch.empty(8, (P + 127) // 128 * 128, 256, device=xyz.device, dtype=torch.float16) if half else torch.empty(8, P, 256, device=xyz.device)
    d_xyz = torch.empty(P, 3, device=xyz.device)
    amax = torch.empty(8, device=xyz.device) if half else torch.empty(8, 64, device=xyz.device)
    _lib.check(lib.hnrf_canonical_bwd(_ptr(xyz), _ptr(d_raw), bits.data_ptr(), _ptr(packed), TRAIN_MODES[mode],
                                      _ptr(d_raw_amax), P, _ptr(dZ), _ptr(d_xyz), _ptr(amax), _stream()), 'hnrf_canonical_bwd')
    return dZ, d_xyz, amax


def nonrigid_bwd(x_skel, hann_w, d_xyz, bits, weights, mode='f32'):
    """dX chain of the non-rigid MLP: returns dZ (6,P,128), d_x_skel (P,3) (identity path included), amax (6,64)
    (mode 'f16x3h': float16 dZ and scale (6,), see canonical_bwd)."""
    lib = _lib.load()
    _chk(x_skel, hann_w, d_xyz, *weights)
    P = x_skel.numel() // 3
    assert bits.shape == (6, P, 4) and bits.dtype == torch.int32 and bits.is_contiguous()
    assert d_xyz.numel() == 3 * P and len(weights) == 7
    half = mode == 'f16x3h'
    m = MLP_MODES['f16x3' if half else mode]
    packed = torch.empty((lib.hnrf_nonrigid_bwd_packed_bytes(m) + 3) // 4, device=x_skel.device)
    _lib.check(lib.hnrf_nonrigid_bwd_pack(_ptr_array(weights), m, _ptr(packed), _stream()), 'hnrf_nonrigid_bwd_pack')
    d_amax = d_xyz.abs().amax().reshape(1) if mode != 'f32' else None
    dZ = torch.empty(6, (P + 127) // 128 * 128, 128, device=x_skel.device, dtype=torch.float16) if half else torch.empty(6, P, 128, device=x_skel.device)
    d_x_skel = torch.empty(P, 3, device=x_skel.device)
    amax = torch.empty(6, device=x_skel.device) if half else torch.empty(6, 64, device=x_skel.device)
    _lib.check(lib.hnrf_nonrigid_bwd(_ptr(x_skel), _ptr(hann_w), _ptr(d_xyz), bits.data_ptr(), _ptr(packed),
                                     TRAIN_MODES[mode], _ptr(d_amax), P, _ptr(dZ), _ptr(d_x_skel), _ptr(amax), _stream()),
               'hnrf_nonrigid_bwd')
    return dZ, d_x_skel, amax


_dw_ws = {}


def mlp_dw(dZ, X, dW_out=None, want_db=True, mode='f32', dz_amax=None):
    """dW = dZ^T X (and db = column sums of dZ) for one nn.Linear: dZ [P, n_out], X [P, n_in], row-major views
    whose last dimension is contiguous.  ``dW_out``: optional [n_out, >= n_in] view to write into (row stride kept).
    mode 'f16x3' (matrix-shaped layers only, others fall back to fp32 MFMA) needs ``dz_amax``: a device scalar
    tensor whose maximum bounds |dZ| (one row of the chain kernels' amax output)."""
    lib = _lib.load()
    assert dZ.dtype == torch.float32 and X.dtype == torch.float32 and dZ.is_cuda and X.is_cuda
    assert dZ.dim() == 2 and X.dim() == 2 and dZ.stride(1) == 1 and X.stride(1) == 1 and dZ.shape[0] == X.shape[0]
    P, n_out = dZ.shape
    n_in = X.shape[1]
    if dW_out is None:
        dW_out = torch.empty(n_out, n_in, device=dZ.device)
    assert dW_out.stride(1) == 1 and dW_out.shape[0] == n_out and dW_out.shape[1] == n_in
    db = torch.empty(n_out, device=dZ.device) if want_db else None
    need = lib.hnrf_mlp_dw_workspace_bytes(P, n_out, n_in)
    if need == 0:
        raise _lib.HnrfError('hnrf_mlp_dw: shape (%d, %d, %d) not built' % (P, n_out, n_in))
    key = dZ.device.index
    ws = _dw_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _dw_ws[key] = torch.empty(need, dtype=torch.uint8, device=dZ.device)
    if mode == 'f16x3' and dz_amax is None:
        dz_amax = dZ.abs().amax().reshape(1)
    _lib.check(lib.hnrf_mlp_dw(dZ.data_ptr(), dZ.stride(0), X.data_ptr(), X.stride(0), P, n_out, n_in,
                               MLP_MODES[mode], _ptr(dz_amax), 0 if dz_amax is None else dz_amax.numel(),
                               dW_out.data_ptr(), dW_out.stride(0), _ptr(db), ws.data_ptr(), ws.numel(), _stream()),
               'hnrf_mlp_dw')
    return dW_out, db


def mlp_dw_h(dZ, X, dW_out=None, want_db=True, dz_scale=None, n_in=None, P=None, z_blocked=False, x_blocked=False):
    """hnrf_mlp_dw_h: dW = dZ^T X / scale (and db) from f16 operands.  dZ [P, n_out] f16 (n_out 128 | 256) or, for a
    head, fp32 [P, n_out <= 4]; X [P, >= n_in] f16 whose rows are zero-padded to 64 / 128 / 256 columns; ``n_in``:
    columns of dW (default X.shape[1]); dz_scale: device scalar the stored dZ was multiplied by.  ``z_blocked`` /
    ``x_blocked``: the matrix is one layer of the blocked buffers canonical_train / canonical_bwd return in mode
    'f16x3h' (shape [P padded to 128, width]); ``P`` then gives the true sample count."""
    lib = _lib.load()
    head = dZ.shape[1] <= 4
    assert X.dtype == torch.float16 and dZ.dtype == (torch.float32 if head else torch.float16) and dZ.is_cuda and X.is_cuda
    assert dZ.dim() == 2 and X.dim() == 2 and dZ.stride(1) == 1 and X.stride(1) == 1
    n_out = dZ.shape[1]
    P = int(dZ.shape[0] if P is None else P)
    assert dZ.shape[0] >= P and X.shape[0] >= P and (z_blocked or x_blocked or dZ.shape[0] == X.shape[0])
    assert (not z_blocked or dZ.is_contiguous()) and (not x_blocked or X.is_contiguous())
    n_in = int(X.shape[1] if n_in is None else n_in)
    if dW_out is None:
        dW_out = torch.empty(n_out, n_in, device=dZ.device)
    assert dW_out.stride(1) == 1 and dW_out.shape[0] == n_out and dW_out.shape[1] == n_in and dW_out.dtype == torch.float32
    db = torch.empty(n_out, device=dZ.device) if want_db else None
    need = lib.hnrf_mlp_dw_h_workspace_bytes(P, n_out, n_in)
    if need == 0:
        raise _lib.HnrfError('hnrf_mlp_dw_h: shape (%d, %d, %d) not built' % (P, n_out, n_in))
    key = dZ.device.index
    ws = _dw_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _dw_ws[key] = torch.empty(need, dtype=torch.uint8, device=dZ.device)
    layout = (1 if z_blocked else 0) | (2 if x_blocked else 0)
    _lib.check(lib.hnrf_mlp_dw_h(dZ.data_ptr(), dZ.stride(0), X.data_ptr(), X.stride(0), P, n_out, n_in, layout,
                                 _ptr(dz_scale), dW_out.data_ptr(), dW_out.stride(0), _ptr(db), ws.data_ptr(), ws.numel(),
                                 _stream()), 'hnrf_mlp_dw_h')
    return dW_out, db


def motion_basis_fwd(dst_Rs, dst_Ts, cnl_gtfms, want_saved=False, rvec=None):
    """hnrf_motion_basis_fwd: (B,3,3), (B,3), (B,4,4) -> Rs (B,3,3), Ts (B,3) [, saved state for the backward].
    With ``rvec`` (B-1,3): hnrf_refined_motion_basis_fwd (the pose refinement's Rodrigues correction folded in)."""
    lib = _lib.load()
    _chk(dst_Rs, dst_Ts, cnl_gtfms)
    B = dst_Rs.shape[0]
    assert dst_Rs.shape == (B, 3, 3) and dst_Ts.shape == (B, 3) and cnl_gtfms.shape == (B, 4, 4)
    Rs, Ts = torch.empty_like(dst_Rs), torch.empty_like(dst_Ts)
    saved = torch.empty(lib.hnrf_motion_basis_saved_bytes() // 8, dtype=torch.float64, device=dst_Rs.device) if want_saved else None
    if rvec is None:
        _lib.check(lib.hnrf_motion_basis_fwd(_ptr(dst_Rs), _ptr(dst_Ts), _ptr(cnl_gtfms), B, _ptr(Rs), _ptr(Ts), _ptr(saved),
                                             _stream()), 'hnrf_motion_basis_fwd')
    else:
        _chk(rvec)
        assert rvec.shape == (B - 1, 3)
        _lib.check(lib.hnrf_refined_motion_basis_fwd(_ptr(rvec), _ptr(dst_Rs), _ptr(dst_Ts), _ptr(cnl_gtfms), B, _ptr(Rs),
                                                     _ptr(Ts), _ptr(saved), _stream()), 'hnrf_refined_motion_basis_fwd')
    return Rs, Ts, saved


def motion_basis_bwd(g_Rs, g_Ts, dst_Rs, dst_Ts, cnl_gtfms, saved, rvec=None):
    """-> d_dst_Rs, d_dst_Ts [, d_rvec when ``rvec`` is given]."""
    lib = _lib.load()
    _chk(g_Rs, g_Ts, dst_Rs, dst_Ts, cnl_gtfms)
    d_Rs, d_Ts = torch.empty_like(dst_Rs), torch.empty_like(dst_Ts)
    if rvec is None:
        _lib.check(lib.hnrf_motion_basis_bwd(_ptr(g_Rs), _ptr(g_Ts), _ptr(dst_Rs), _ptr(dst_Ts), _ptr(cnl_gtfms), dst_Rs.shape[0],
                                             _ptr(saved), _ptr(d_Rs), _ptr(d_Ts), _stream()), 'hnrf_motion_basis_bwd')
        return d_Rs, d_Ts
    _chk(rvec)
    d_rvec = torch.empty_like(rvec)
    _lib.check(lib.hnrf_refined_motion_basis_bwd(_ptr(g_Rs), _ptr(g_Ts), _ptr(rvec), _ptr(dst_Rs), _ptr(dst_Ts), _ptr(cnl_gtfms),
                                                 dst_Rs.shape[0], _ptr(saved), _ptr(d_rvec), _ptr(d_Rs), _ptr(d_Ts), _stream()),
               'hnrf_refined_motion_basis_bwd')
    return d_Rs, d_Ts, d_rvec


def _pose_mlp_dims(weights):
    import ctypes
    dims = [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]
    for l, w in enumerate(weights):
        assert tuple(w.shape) == (dims[l + 1], dims[l]), 'layer %d: %s' % (l, tuple(w.shape))
    return (ctypes.c_int * len(dims))(*dims), dims


def pose_mlp_fwd(x, weights, biases):
    """hnrf_pose_mlp_fwd: x (n_in,) through Linear+ReLU ... Linear -> (n_out,), saved state for the backward."""
    lib = _lib.load()
    _chk(x, *weights, *biases)
    cdims, dims = _pose_mlp_dims(weights)
    L = len(weights)
    out = torch.empty(dims[-1], device=x.device)
    saved = torch.empty(lib.hnrf_pose_mlp_saved_bytes(L) // 4, device=x.device)
    _lib.check(lib.hnrf_pose_mlp_fwd(_ptr(x), _ptr_array(weights), _ptr_array(biases), cdims, L, _ptr(out), _ptr(saved),
                                     _stream()), 'hnrf_pose_mlp_fwd')
    return out, saved


def pose_mlp_bwd(g_out, x, weights, biases, saved, want_dx=False):
    """-> [dW_l], [db_l], d_x or None."""
    lib = _lib.load()
    _chk(g_out, x, saved, *weights, *biases)
    cdims, dims = _pose_mlp_dims(weights)
    dW = [torch.empty_like(w) for w in weights]
    db = [torch.empty_like(b) for b in biases]
    parts = torch.empty(8, 256, device=x.device) if want_dx else None
    _lib.check(lib.hnrf_pose_mlp_bwd(_ptr(g_out), _ptr(x), _ptr_array(weights), _ptr_array(biases), cdims, len(weights),
                                     _ptr(saved), _ptr_array(dW), _ptr_array(db), _ptr(parts), _stream()), 'hnrf_pose_mlp_bwd')
    d_x = parts[:(dims[1] + 31) // 32, :dims[0]].sum(0) if want_dx else None
    return dW, db, d_x


def sample_warp_bwd(rays_o, rays_d, z_vals, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, x_skel, fg_mask,
                    g_x_skel, g_mask):
    """Returns d_vol (same shape as vol; background channel zero), d_Rs (B,3,3), d_Ts (B,3)."""
    lib = _lib.load()
    _chk(rays_o, rays_d, z_vals, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, x_skel, fg_mask, g_x_skel, g_mask)
    R, S = z_vals.shape
    B, G = motion_Rs.shape[0], vol.shape[-1]
    d_vol = torch.zeros_like(vol)
    d_Rs, d_Ts = torch.empty_like(motion_Rs), torch.empty_like(motion_Ts)
    _lib.check(lib.hnrf_sample_warp_bwd(_ptr(rays_o), _ptr(rays_d), _ptr(z_vals), _ptr(motion_Rs), _ptr(motion_Ts),
                                        _ptr(vol), _ptr(bbox_min), _ptr(bbox_scale), _ptr(x_skel), _ptr(fg_mask),
                                        _ptr(g_x_skel), _ptr(g_mask), R, S, B, G, _ptr(d_vol), _ptr(d_Rs), _ptr(d_Ts),
                                        _stream()), 'hnrf_sample_warp_bwd')
    return d_vol, d_Rs, d_Ts


# ----------------------------------------------------------------------------- sample culling
def compact_samples(fg_mask, eps):
    """idx (P,) int32 and count (1,) int32 (device): samples with fg_mask >= eps."""
    lib = _lib.load()
    _chk(fg_mask)
    P = fg_mask.numel()
    idx = torch.empty(P, dtype=torch.int32, device=fg_mask.device)
    count = torch.empty(1, dtype=torch.int32, device=fg_mask.device)
    _lib.check(lib.hnrf_compact_samples(_ptr(fg_mask), float(eps), P, _ptr(idx), _ptr(count), _stream()),
               'hnrf_compact_samples')
    return idx, count


def canonical_sparse(xyz, packed, idx, count, mode='f32', raw=None):
    lib = _lib.load()
    _chk(xyz, packed)
    P = xyz.numel() // 3
    if raw is None:
        raw = torch.zeros(*xyz.shape[:-1], 4, device=xyz.device)
    _lib.check(lib.hnrf_canonical_fwd_sparse(_ptr(xyz), _ptr(packed), _mode_arg(mode), P, _ptr(idx), _ptr(count),
                                             _ptr(raw), _stream()), 'hnrf_canonical_fwd_sparse')
    return raw


def nonrigid_sparse(x_skel, hann_w, packed, idx, count, mode='f32'):
    lib = _lib.load()
    _chk(x_skel, hann_w, packed)
    P = x_skel.numel() // 3
    xyz = x_skel.clone()
    _lib.check(lib.hnrf_nonrigid_fwd_sparse(_ptr(x_skel), _ptr(hann_w), _ptr(packed), _mode_arg(mode), P, _ptr(idx),
                                            _ptr(count), _ptr(xyz), 0, _stream()), 'hnrf_nonrigid_fwd_sparse')
    return xyz


def gen_rays(K, E, bbox_min, bbox_max, H, W, device=None):
    """Rays of one camera that cross the bbox, generated on the device in pixel order.

    K (3,3), E (4,4) float32 arrays / tensors (intrinsics, world->camera extrinsics), bbox_min / bbox_max (3,).
    Returns dict(rays (3,N,3) = [o, d, d], near (N,1), far (N,1), ray_mask (H*W,) bool) -- the per-frame entries a
    reference dataset yields (freeview.py:232-242).  One host sync (the ray count sizes the views)."""
    lib = _lib.load()
    import numpy as np
    device = device or torch.device('cuda', torch.cuda.current_device())
    K = np.asarray(K.cpu() if torch.is_tensor(K) else K, dtype=np.float32)
    E = np.asarray(E.cpu() if torch.is_tensor(E) else E, dtype=np.float32)
    small = np.concatenate([np.linalg.inv(K).reshape(-1), E[:3, :3].reshape(-1), E[:3, 3].reshape(-1),
                            np.asarray(bbox_min, np.float32).reshape(-1), np.asarray(bbox_max, np.float32).reshape(-1)])
    cam = torch.from_numpy(small.astype(np.float32)).to(device)
    n = H * W
    rays_o, rays_d = torch.empty(n, 3, device=device), torch.empty(n, 3, device=device)
    near, far = torch.empty(n, device=device), torch.empty(n, device=device)
    mask = torch.empty(n, dtype=torch.uint8, device=device)
    count = torch.empty(1, dtype=torch.int32, device=device)
    ws = torch.empty(lib.hnrf_gen_rays_workspace_bytes(H, W), dtype=torch.uint8, device=device)
    base = cam.data_ptr()
    _lib.check(lib.hnrf_gen_rays(base, base + 36, base + 72, base + 84, base + 96, H, W, _ptr(rays_o), _ptr(rays_d),
                                 _ptr(near), _ptr(far), mask.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _stream()), 'hnrf_gen_rays')
    N = int(count.item())
    o, d = rays_o[:N], rays_d[:N]
    return {'rays': torch.stack([o, d, d], 0), 'near': near[:N, None], 'far': far[:N, None], 'ray_mask': mask.bool()}


# ------------------------------------------------------------------------------------------------ image pre-processing
# (hnrf_image.hip; the host statement of the same OpenCV functions is imageproc.py)
_image_consts = {}                    # (kind, key..., device) -> device tensors built once per camera / image size


def _upload(arr, device):
    """Small host array -> device through pinned memory (a pageable copy would wait for the stream's queue)."""
    t = torch.from_numpy(arr)
    return t.pin_memory().to(device, non_blocking=True) if torch.device(device).type == 'cuda' else t.to(device)


def _u8_image(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.dim() == 3):
        raise _lib.HnrfError('%s must be a contiguous uint8 (H, W, C) tensor on the GPU' % what)


def undistort_image(img, K, D):
    """cv2.undistort(img, K, D) (train.py:366-371) of a uint8 (H, W, C) image on the device -> new tensor.  K (3,3),
    D (5,) host arrays; the per-camera constants (stripe inverses) are uploaded once per camera and image size."""
    import numpy as np
    from . import imageproc
    lib = _lib.load()
    _u8_image(img, 'undistort_image: img')
    H, W, C = img.shape
    A = np.asarray(K, dtype=np.float64)[:3, :3]
    d = imageproc.distortion_vector(D)
    key = ('undistort', A.tobytes(), d.tobytes(), H, W, img.device)
    c = _image_consts.get(key)
    if c is None:
        rows, ir = imageproc.undistort_stripes(A, H, W)
        cam = np.concatenate([[A[0, 0], A[1, 1], A[0, 2], A[1, 2]], d])
        c = _image_consts[key] = (_upload(cam, img.device), _upload(np.ascontiguousarray(ir), img.device), len(ir), rows)
    out = torch.empty_like(img)
    _lib.check(lib.hnrf_undistort_image(img.data_ptr(), H, W, C, c[0].data_ptr(), c[1].data_ptr(), c[2], c[3],
                                        out.data_ptr(), _stream()), 'hnrf_undistort_image')
    return out


def _resize_tables(Hs, Ws, scale, kind, device):
    import numpy as np
    from . import imageproc
    key = ('tables', Hs, Ws, float(scale), kind, device)
    t = _image_consts.get(key)
    if t is None:
        Hd, Wd = imageproc.resized_size(Hs, Ws, scale)
        inv = 1.0 / float(scale)
        xo, xw = imageproc.resize_tables(Ws, Wd, inv, kind)
        yo, yw = imageproc.resize_tables(Hs, Hd, inv, kind)
        t = _image_consts[key] = (Hd, Wd) + tuple(_upload(np.ascontiguousarray(a), device) for a in (xo, xw, yo, yw))
    return t


def composite_windows(orig, alpha, bgcolor, windows, ph, pw, scale=1.0):
    """float32 [n, ph, pw, 3] = pixels of ``load_image(...)[0] / 255`` (train.py:406-417: composite over ``bgcolor``
    (0..255, float32 [3] on the device), then INTER_LANCZOS4 to ``scale``) inside the n windows whose top-left
    destination pixels are ``windows`` ((n, 2) of (x0, y0): host ints, checked here).  orig / alpha: uint8 (Hs, Ws, 3)
    on the device, already undistorted.  One whole image = one window at (0, 0)."""
    import numpy as np
    from . import imageproc
    lib = _lib.load()
    _u8_image(orig, 'composite_windows: orig')
    _u8_image(alpha, 'composite_windows: alpha')
    _chk(bgcolor)
    Hs, Ws, C = orig.shape
    assert C == 3 and alpha.shape == orig.shape and bgcolor.numel() == 3
    win = np.ascontiguousarray(np.asarray(windows, dtype=np.int32).reshape(-1, 2))
    resize = 0 if float(scale) == 1.0 else 1
    if resize:
        Hd, Wd, xo, xw, yo, yw = _resize_tables(Hs, Ws, scale, 'lanczos4', orig.device)
    else:
        Hd, Wd, xo, xw, yo, yw = Hs, Ws, None, None, None, None
    if len(win) == 0 or win.min() < 0 or (win[:, 0] + pw).max() > Wd or (win[:, 1] + ph).max() > Hd:
        raise _lib.HnrfError('composite_windows: windows %s of %dx%d leave the %dx%d image' % (win.tolist(), ph, pw, Hd, Wd))
    win_d = _upload(win, orig.device)
    out = torch.empty(len(win), ph, pw, 3, device=orig.device)
    _lib.check(lib.hnrf_composite_windows(orig.data_ptr(), alpha.data_ptr(), Hs, Ws, bgcolor.data_ptr(), resize, _ptr(xo),
                                          _ptr(xw), _ptr(yo), _ptr(yw), Hd, Wd, win_d.data_ptr(), len(win), ph, pw,
                                          out.data_ptr(), _stream()), 'hnrf_composite_windows')
    return out


def resize_mask(alpha, scale, channel=0):
    """float32 (Hd, Wd): channel ``channel`` of ``cv2.resize(alpha / 255, fx=scale, fy=scale, INTER_LINEAR)``
    (train.py:413-417); at scale 1 the channel / 255 itself."""
    from . import imageproc
    lib = _lib.load()
    _u8_image(alpha, 'resize_mask: alpha')
    Hs, Ws, C = alpha.shape
    assert C == 3
    if float(scale) == 1.0:
        return (alpha[:, :, channel].double() / 255.).float()
    Hd, Wd = imageproc.resized_size(Hs, Ws, scale)
    out = torch.empty(Hd, Wd, device=alpha.device)
    if imageproc.is_half_scale(scale) and 2 * Hd <= Hs and 2 * Wd <= Ws:
        _lib.check(lib.hnrf_resize_mask(alpha.data_ptr(), Hs, Ws, channel, 2, 0, 0, 0, 0, Hd, Wd, out.data_ptr(), _stream()),
                   'hnrf_resize_mask')
    else:
        _, _, xo, xw, yo, yw = _resize_tables(Hs, Ws, scale, 'linear', alpha.device)
        _lib.check(lib.hnrf_resize_mask(alpha.data_ptr(), Hs, Ws, channel, 1, xo.data_ptr(), xw.data_ptr(), yo.data_ptr(),
                                        yw.data_ptr(), Hd, Wd, out.data_ptr(), _stream()), 'hnrf_resize_mask')
    return out


def deconv_fold(col, bias, cout, D, H, W):
    """hnrf_deconv_fold: col (D*H*W, cout*64) fp32 -> (1, cout, 2D, 2H, 2W): the fold of ConvTranspose3d(4, 2, 1)."""
    lib = _lib.load()
    _chk(col, bias)
    assert tuple(col.shape) == (D * H * W, cout * 64) and (bias is None or bias.numel() == cout)
    out = torch.empty(1, cout, 2 * D, 2 * H, 2 * W, device=col.device)
    _lib.check(lib.hnrf_deconv_fold(_ptr(col), _ptr(bias), cout, D, H, W, _ptr(out), _stream()), 'hnrf_deconv_fold')
    return out
