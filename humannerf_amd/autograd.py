"""torch.autograd glue for training through the HIP path.

``RenderRays`` is one differentiable op = Network._render_rays (network.py:474-602) for one
ray chunk.  Forward: K1, K2/K3 in their activation-saving variants (cfg.amd.train_mlp_mode), K4.  Backward: K4',
for the two MLPs the register-resident dX chain (hnrf_*_bwd: relu' from the saved
post-activations, PE' fused) and hnrf_mlp_dw (dW = dZ^T X and db, one workgroup per CU
holding the whole output), K1'.  No library GEMM is left on the per-sample path.  Gradients are produced for the motion
bases, the weight volume and all MLP parameters -- exactly the tensors through which the
reference's four parameter groups receive gradient (last row).
"""
import math

import torch

from . import ops
from .config import amd_option


def _weight_grads(dZ, acts, pe, dY, weights, skip_layer, skip_order, amax=None, mode='f32'):
    """dW / db of every layer from the chain kernel's dZ [L][P][W], the saved activations and PE matrix.
    dY: gradient at the (3- / 4-wide) head output.  mode 'f16x3' (with amax from the chain kernel) runs the
    matrix-shaped layers on the split-f16 kernel; PE blocks and heads stay fp32."""
    n_hidden = acts.shape[0]
    npe = pe.shape[1]
    gW, gb = [None] * (n_hidden + 1), [None] * (n_hidden + 1)
    am = (lambda l: amax[l]) if amax is not None else (lambda l: None)
    if amax is None:
        mode = 'f32'
    gW[n_hidden], gb[n_hidden] = ops.mlp_dw(dY, acts[n_hidden - 1])
    for l in range(n_hidden):
        if l == 0:
            gW[l], gb[l] = ops.mlp_dw(dZ[l], pe)
        elif l == skip_layer:
            gW[l] = torch.empty_like(weights[l])
            pe_cols = gW[l][:, :npe] if skip_order == 'pe_first' else gW[l][:, -npe:]
            h_cols = gW[l][:, npe:] if skip_order == 'pe_first' else gW[l][:, :-npe]
            ops.mlp_dw(dZ[l], pe, pe_cols, want_db=False)
            _, gb[l] = ops.mlp_dw(dZ[l], acts[l - 1], h_cols, mode=mode, dz_amax=am(l))
        else:
            gW[l], gb[l] = ops.mlp_dw(dZ[l], acts[l - 1], mode=mode, dz_amax=am(l))
    return gW, gb


def _weight_grads_h(dZ, scale, acts, pe, dY, weights, skip_layer, skip_order, npe):
    """The same from f16 operands (cfg.amd.train_operands = 'f16'): dZ [L][P128][W] f16 in the chain's scaled domain
    with ``scale`` [L], acts f16 -- both in the blocked layout the training kernels write -- pe row-major f16 [P][64]
    (``npe`` real columns), dY fp32 [P, 3|4] at the head output.  Every layer runs on hnrf_mlp_dw_h (transposed LDS reads,
    one f16 MFMA per product, fp32 accumulation)."""
    n_hidden = acts.shape[0]
    P = dY.shape[0]
    gW, gb = [None] * (n_hidden + 1), [None] * (n_hidden + 1)
    gW[n_hidden], gb[n_hidden] = ops.mlp_dw_h(dY, acts[n_hidden - 1], P=P, x_blocked=True)
    for l in range(n_hidden):
        sc = scale[l:l + 1]
        if l == 0:
            gW[l], gb[l] = ops.mlp_dw_h(dZ[l], pe, dz_scale=sc, n_in=npe, P=P, z_blocked=True)
        elif l == skip_layer:
            gW[l] = torch.empty_like(weights[l])
            pe_cols = gW[l][:, :npe] if skip_order == 'pe_first' else gW[l][:, -npe:]
            h_cols = gW[l][:, npe:] if skip_order == 'pe_first' else gW[l][:, :-npe]
            ops.mlp_dw_h(dZ[l], pe, pe_cols, want_db=False, dz_scale=sc, n_in=npe, P=P, z_blocked=True)
            _, gb[l] = ops.mlp_dw_h(dZ[l], acts[l - 1], h_cols, dz_scale=sc, P=P, z_blocked=True, x_blocked=True)
        else:
            gW[l], gb[l] = ops.mlp_dw_h(dZ[l], acts[l - 1], dz_scale=sc, P=P, z_blocked=True, x_blocked=True)
    return gW, gb


ACT_MIN, ACT_MAX = 2.0 ** -8, 60000.0


class OperandRangeGuard:
    """Run-time check of the premise of the f16 / split-f16 training arithmetic: every layer's largest
    activation must sit inside f16's useful range -- below ~2^-8 the operands of the weight-gradient products lose
    their bits to f16 subnormals, at 65504 the kernels clamp -- and the incoming gradient must be finite.  Checked every
    ``cfg.amd.train_check_every`` backward passes (one reduction over the saved activations, ~0.5 ms); the verdict is
    read back without a synchronisation and raised at the next check.  Remedy when it fires: cfg.amd.train_mlp_mode =
    train_chain_mode = train_dw_mode = 'f32' (exact fp32 MFMA kernels)."""

    def __init__(self):
        self.calls = 0
        self.pending = []

    @staticmethod
    def flags(acts_list, d_in, dz_list=()):
        """device bool tensor [too_small, too_large, non_finite_gradient]; dz_list: f16 dZ buffers of the chains (their
        scaled values must stay below the f16 limit: the chain clamps there)"""
        def layer_amax(a):                           # two-stage: a reduction to L outputs alone runs on L workgroups
            lo, hi = torch.aminmax(a.reshape(a.shape[0], math.gcd(a[0].numel(), 1024), -1), dim=2)
            return torch.maximum(hi.amax(dim=1), -lo.amin(dim=1)).float()
        amax = torch.cat([layer_amax(a) for a in acts_list if a is not None])
        large = (amax >= ACT_MAX).any()
        for dz in dz_list:
            if dz is not None and dz.dtype == torch.float16:
                large = large | (layer_amax(dz) >= ACT_MAX).any()
        fin = torch.isfinite(d_in).all()
        # (a layer whose activations are ALL exactly zero -- a ReLU layer fed zeros with zero biases, as at the start of a
        # run -- has nothing to lose in f16)
        return torch.stack([((amax < ACT_MIN) & (amax > 0)).any(), large, ~fin])

    def due(self):
        every = int(amd_option('train_check_every', 200))
        return every > 0 and (self.calls - 1) % every == 0

    def maybe_check(self, acts_list, d_in, dz_list=()):
        self.poll()
        if not self.due():
            return
        fl = self.flags(acts_list, d_in, dz_list)
        host = torch.empty(3, dtype=torch.bool, pin_memory=True)
        host.copy_(fl, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((host, ev, self.calls))

    def poll(self, wait=False):
        keep = []
        for host, ev, call in self.pending:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((host, ev, call))
                continue
            small, large, nonfinite = (bool(x) for x in host)
            if small or large or nonfinite:
                self.pending = []
                raise ops._lib.HnrfError(
                    'training operands left the range of the split-f16 arithmetic at backward pass %d: %s.  Set '
                    "cfg.amd.train_mlp_mode = train_chain_mode = train_dw_mode = 'f32' for this model." % (
                        call, ', '.join(n for n, f in (('a layer with all activations below 2^-8', small),
                                                       ('activations or scaled gradients at the f16 limit (clamped)', large),
                                                       ('non-finite incoming gradient', nonfinite)) if f)))
        self.pending = keep


range_guard = OperandRangeGuard()


def training_modes():
    """(forward, chain, dW) arithmetic of the training kernels and whether the saved operands are f16.
    cfg.amd.train_operands = 'f16' needs split-f16 arithmetic in the forward and the chain."""
    fwd, chain, dw = (amd_option('train_mlp_mode', 'f16x3'), amd_option('train_chain_mode', 'f16x3'),
                      amd_option('train_dw_mode', 'f16x3'))
    half = amd_option('train_operands', 'f16') == 'f16' and fwd == 'f16x3' and chain == 'f16x3' and dw == 'f16x3'
    return fwd, chain, dw, half


class RenderRays(torch.autograd.Function):
    """rgb, alpha, depth[, 8 diagnostic outputs] = RenderRays.apply(consts..., motion_Rs, motion_Ts, vol, *mlp_params)

    With ``diag`` the other keys of the reference's return dict (network.py:776-789) come out too, as
    non-differentiable tensors, in OUTPUT_KEYS order."""

    OUTPUT_KEYS = ('rgb', 'alpha', 'depth', 'weights_on_rays', 'rgb_on_rays', 'cnl_xyz', 'cnl_rgb', 'cnl_weight',
                   'xyz_on_rays', 'backward_motion_weights', 'offsets')

    @staticmethod
    def forward(ctx, rays_o, rays_d, near, far, t_rand, bbox_min, bbox_scale, hann_w, cond, bg, n_samples,
                use_nonrigid, diag, const_offset, motion_Rs, motion_Ts, vol, *params):
        """``const_offset`` (3,) or None: the non-rigid offset when it is ONE vector for the whole frame -- before
        non_rigid_motion_mlp.kick_in_iter the condition code and every Hann weight are zero (network.py:735-737,
        hannw_fourier.py:28-40), so the MLP sees zeros at every sample and returns MLP(0).  The caller evaluates that once
        (differentiably, in fp32) and passes ``use_nonrigid = False``: xyz = x_skel + const_offset."""
        nr_w, nr_b = list(params[0:7]), list(params[7:14])
        cn_w, cn_b = list(params[14:23]), list(params[23:32])
        motion_Rs, motion_Ts, vol = motion_Rs.contiguous(), motion_Ts.contiguous(), vol.contiguous()
        z, x_skel, mask, bmw = ops.sample_warp(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min,
                                               bbox_scale, n_samples, want_bmw=bool(diag))
        mode, _, _, half = training_modes()
        tmode = 'f16x3h' if half else mode
        if use_nonrigid:
            nr_packed = ops.nonrigid_pack(nr_w, nr_b, cond, mode)
            xyz, offsets, pe_n, acts_n, bits_n = ops.nonrigid_train(x_skel, hann_w, nr_packed, tmode)
        elif const_offset is not None:
            xyz, pe_n, acts_n, bits_n = (x_skel + const_offset.detach().reshape(1, 1, 3)).contiguous(), None, None, None
            offsets = const_offset.detach().reshape(1, 1, 3).expand_as(x_skel).contiguous() if diag else None
        else:
            xyz, pe_n, acts_n, bits_n = x_skel, None, None, None
            offsets = torch.zeros_like(x_skel) if diag else None            # network.py:276-277
        cn_packed = ops.canonical_pack(cn_w, cn_b, mode)
        raw, pe_c, acts_c, bits_c = ops.canonical_train(xyz, cn_packed, tmode)
        out = ops.composite(raw, mask, z, rays_d, xyz if diag else None, bg, diagnostics=bool(diag))
        ctx.use_nonrigid = use_nonrigid
        ctx.const_offset = const_offset is not None
        ctx.half = half
        ctx.n_out = 11 if diag else 3
        ctx.save_for_backward(rays_o, rays_d, z, x_skel, mask, xyz, raw, pe_c, acts_c, pe_n, acts_n, motion_Rs,
                              motion_Ts, vol, bbox_min, bbox_scale, hann_w, cond, bg, bits_c, bits_n, *nr_w, *cn_w)
        if not diag:
            return out['rgb'], out['alpha'], out['depth']
        out.update(xyz_on_rays=xyz.view_as(x_skel).clone(), backward_motion_weights=bmw, offsets=offsets)
        extra = tuple(out[k] for k in RenderRays.OUTPUT_KEYS[3:])
        ctx.mark_non_differentiable(*extra)
        return (out['rgb'], out['alpha'], out['depth']) + extra

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, *_unused):
        (rays_o, rays_d, z, x_skel, mask, xyz, raw, pe_c, acts_c, pe_n, acts_n, motion_Rs, motion_Ts, vol, bbox_min,
         bbox_scale, hann_w, cond, bg, bits_c, bits_n) = ctx.saved_tensors[:21]
        nr_w = list(ctx.saved_tensors[21:28])
        cn_w = list(ctx.saved_tensors[28:37])
        P = z.numel()
        c = lambda t: None if t is None else t.contiguous()
        d_raw, d_mask = ops.composite_bwd(raw, mask, z, rays_d, bg, c(g_rgb), c(g_alpha), c(g_depth))
        # canonical MLP (skip layer 5 takes [PE63 | h]): dX chain with the PE backward fused, then the weight gradients
        d_raw = d_raw.view(P, 4)
        _, chain_mode, dw_mode, _ = training_modes()
        guard = chain_mode != 'f32' or dw_mode != 'f32'
        if guard:
            range_guard.calls += 1
        if ctx.half:
            dZc, d_xyz, sc_c = ops.canonical_bwd(xyz.reshape(P, 3), d_raw, bits_c, cn_w, 'f16x3h')
            gWc, gbc = _weight_grads_h(dZc, sc_c, acts_c, pe_c, d_raw, cn_w, skip_layer=5, skip_order='pe_first', npe=63)
        else:
            dZc, d_xyz, amax_c = ops.canonical_bwd(xyz.reshape(P, 3), d_raw, bits_c, cn_w, chain_mode)
            gWc, gbc = _weight_grads(dZc, acts_c, pe_c, d_raw, cn_w, skip_layer=5, skip_order='pe_first', amax=amax_c,
                                     mode=dw_mode)
        dZc_keep = dZc if (guard and range_guard.due()) else None
        del dZc
        dZn_keep = None
        if ctx.use_nonrigid:
            # xyz = x_skel + offset; layer 0 input [cond69 | PE36], skip layer 4 takes [h | PE36]
            if ctx.half:
                dZn, d_x_skel, sc_n = ops.nonrigid_bwd(x_skel.reshape(P, 3), hann_w, d_xyz, bits_n, nr_w, 'f16x3h')
                gWn, gbn = _weight_grads_h(dZn, sc_n, acts_n, pe_n, d_xyz, nr_w, skip_layer=4, skip_order='h_first', npe=36)
            else:
                dZn, d_x_skel, amax_n = ops.nonrigid_bwd(x_skel.reshape(P, 3), hann_w, d_xyz, bits_n, nr_w, chain_mode)
                gWn, gbn = _weight_grads(dZn, acts_n, pe_n, d_xyz, nr_w, skip_layer=4, skip_order='h_first', amax=amax_n,
                                         mode=dw_mode)
            dZn_keep = dZn if (guard and range_guard.due()) else None
            del dZn
            # condition-code columns of layer 0: the same vector for every sample
            gWn[0] = torch.cat([gbn[0][:, None] * cond.reshape(1, -1), gWn[0]], dim=1)
        else:
            gWn, gbn = [None] * 7, [None] * 7
            d_x_skel = d_xyz
        if guard:           # (before the dZ buffers are released)
            range_guard.maybe_check([acts_c, acts_n], d_raw, [dZc_keep, dZn_keep])
        del dZc_keep, dZn_keep
        d_vol, d_Rs, d_Ts = ops.sample_warp_bwd(rays_o, rays_d, z, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                                                x_skel, mask, d_x_skel.view_as(x_skel).contiguous(), d_mask)
        d_offset = d_xyz.reshape(P, 3).sum(dim=0) if ctx.const_offset else None
        return (None,) * 13 + (d_offset, d_Rs, d_Ts, d_vol, *gWn, *gbn, *gWc, *gbc)
