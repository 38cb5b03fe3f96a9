"""torch.autograd glue for training through the HIP path.

``RenderRays`` is one differentiable op = Network._render_rays (network.py:474-602) for one
ray chunk.  Forward: K1, K2/K3 in their activation-saving variants, K4.  Backward: K4',
for the two MLPs hnrf_mlp_dw (dW = dZ^T X and db, one workgroup per CU holding the whole
output) + library GEMMs for dX = dZ W (relu' from the saved post-activation), PE', K1'.  Gradients are produced for the motion
bases, the weight volume and all MLP parameters -- exactly the tensors through which the
reference's four parameter groups receive gradient (last row).
"""
import torch

from . import ops


def _mlp_backward(dY, acts, weights, first_inputs, skip_layer, skip_order):
    """Backward of a ReLU MLP given the saved post-ReLU activations.

    dY: gradient at the (linear) head output.  acts[l] = output of hidden layer l.
    first_inputs: PE matrix feeding layer 0 (and the skip layer).  skip_order: 'pe_first' |
    'h_first' (column order of the skip layer's weight).  Returns (grad weights, grad biases,
    dPE accumulated over layer 0 and the skip layer)."""
    n_hidden = acts.shape[0]
    gW, gb = [None] * (n_hidden + 1), [None] * (n_hidden + 1)
    pe = first_inputs
    npe = pe.shape[1]
    gW[n_hidden] = dY.t() @ acts[n_hidden - 1]                         # 3- / 4-row head: library GEMM
    gb[n_hidden] = dY.sum(0)
    dH = dY @ weights[n_hidden]
    dPE = None
    for l in range(n_hidden - 1, -1, -1):
        dZ = torch.ops.aten.threshold_backward(dH, acts[l], 0.0)      # relu'
        W = weights[l]
        if l == 0:
            gW[l], gb[l] = ops.mlp_dw(dZ, pe)
            d = dZ @ W[:, -npe:] if W.shape[1] != npe else dZ @ W
            dPE = d if dPE is None else dPE + d
        elif l == skip_layer:
            X = acts[l - 1]
            gW[l] = torch.empty_like(W)
            if skip_order == 'pe_first':
                ops.mlp_dw(dZ, pe, gW[l][:, :npe], want_db=False)
                _, gb[l] = ops.mlp_dw(dZ, X, gW[l][:, npe:])
                dX = dZ @ W
                dPE = dX[:, :npe].contiguous() if dPE is None else dPE + dX[:, :npe]
                dH = dX[:, npe:].contiguous()
            else:
                _, gb[l] = ops.mlp_dw(dZ, X, gW[l][:, :-npe])
                ops.mlp_dw(dZ, pe, gW[l][:, -npe:], want_db=False)
                dX = dZ @ W
                dPE = dX[:, -npe:].contiguous() if dPE is None else dPE + dX[:, -npe:]
                dH = dX[:, :-npe].contiguous()
        else:
            gW[l], gb[l] = ops.mlp_dw(dZ, acts[l - 1])
            dH = dZ @ W
    return gW, gb, dPE


class RenderRays(torch.autograd.Function):
    """rgb, alpha, depth = RenderRays.apply(consts..., motion_Rs, motion_Ts, vol, *mlp_params)"""

    @staticmethod
    def forward(ctx, rays_o, rays_d, near, far, t_rand, bbox_min, bbox_scale, hann_w, cond, bg, n_samples,
                use_nonrigid, motion_Rs, motion_Ts, vol, *params):
        nr_w, nr_b = list(params[0:7]), list(params[7:14])
        cn_w, cn_b = list(params[14:23]), list(params[23:32])
        motion_Rs, motion_Ts, vol = motion_Rs.contiguous(), motion_Ts.contiguous(), vol.contiguous()
        z, x_skel, mask, _ = ops.sample_warp(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min,
                                             bbox_scale, n_samples, want_bmw=False)
        if use_nonrigid:
            nr_packed = ops.nonrigid_pack(nr_w, nr_b, cond, 'f32')
            xyz, _, pe_n, acts_n = ops.nonrigid_train(x_skel, hann_w, nr_packed)
        else:
            xyz, pe_n, acts_n = x_skel, None, None
        cn_packed = ops.canonical_pack(cn_w, cn_b, 'f32')
        raw, pe_c, acts_c = ops.canonical_train(xyz, cn_packed)
        out = ops.composite(raw, mask, z, rays_d, None, bg, diagnostics=False)
        ctx.use_nonrigid = use_nonrigid
        ctx.save_for_backward(rays_o, rays_d, z, x_skel, mask, xyz, raw, pe_c, acts_c, pe_n, acts_n, motion_Rs,
                              motion_Ts, vol, bbox_min, bbox_scale, hann_w, cond, bg, *nr_w, *cn_w)
        return out['rgb'], out['alpha'], out['depth']

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        (rays_o, rays_d, z, x_skel, mask, xyz, raw, pe_c, acts_c, pe_n, acts_n, motion_Rs, motion_Ts, vol, bbox_min,
         bbox_scale, hann_w, cond, bg) = ctx.saved_tensors[:19]
        nr_w = list(ctx.saved_tensors[19:26])
        cn_w = list(ctx.saved_tensors[26:35])
        P = z.numel()
        c = lambda t: None if t is None else t.contiguous()
        d_raw, d_mask = ops.composite_bwd(raw, mask, z, rays_d, bg, c(g_rgb), c(g_alpha), c(g_depth))
        # canonical MLP: skip layer 5 takes [PE63 | h]
        gWc, gbc, dPE = _mlp_backward(d_raw.view(P, 4), acts_c, cn_w, pe_c, skip_layer=5, skip_order='pe_first')
        d_xyz = ops.pe_bwd(xyz.reshape(P, 3), dPE.contiguous(), None, 10, True)
        if ctx.use_nonrigid:
            # xyz = x_skel + offset; layer 0 input [cond69 | PE36], skip layer 4 takes [h | PE36]
            gWn, gbn, dPEn = _mlp_backward(d_xyz, acts_n, nr_w, pe_n, skip_layer=4, skip_order='h_first')
            # condition-code columns of layer 0: the same vector for every sample
            gWn[0] = torch.cat([gbn[0][:, None] * cond.reshape(1, -1), gWn[0]], dim=1)
            d_x_skel = ops.pe_bwd(x_skel.reshape(P, 3), dPEn.contiguous(), hann_w, 6, False, out=d_xyz.clone())
        else:
            gWn, gbn = [None] * 7, [None] * 7
            d_x_skel = d_xyz
        d_vol, d_Rs, d_Ts = ops.sample_warp_bwd(rays_o, rays_d, z, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                                                x_skel, mask, d_x_skel.view_as(x_skel).contiguous(), d_mask)
        return (None,) * 12 + (d_Rs, d_Ts, d_vol, *gWn, *gbn, *gWc, *gbc)
