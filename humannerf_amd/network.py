"""``Network`` -- drop-in for the reference's core/nets/human_nerf/network.py.

Same constructor (no arguments, reads the global ``cfg``), same ``forward``
keyword interface and output keys, same ``state_dict`` key names (including
the ``.module.`` infix nn.DataParallel inserted in the reference -- checkpoint
and optimizer-group compatibility), same attribute names the
optimizer routes learning rates by (optimizer.py:9-34).  Select it with
``network_module: 'humannerf_amd.network'`` (create_network.py:6-15 loads the
dotted path with imp.load_source).

What differs is everything underneath: the per-sample work (sampling, LBS warp,
both MLPs, compositing) runs in the hand-written HIP kernels of libhnrf.so, one
process per GPU, whole renderer replicated on every GPU.  The reference's
primary/secondary GPU split and its per-call parameter broadcast
(network.py:68-72,115-119) do not exist here; ``deploy_mlps_to_secondary_gpus``
is kept as a no-op.  Per-frame work that is a few KFLOP (pose refinement,
kinematic chain, 4x4 inverses) and the once-per-frame weight-volume decoder stay
in PyTorch-ROCm.

Only the default-config branches are implemented; anything else raises.
"""
import math
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .autograd import RenderRays
from .config import cfg, amd_option

SMPL_PARENT = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16,
               17, 18, 19, 20, 21]          # core/utils/network_util.py:91-94


# --------------------------------------------------------------------------- init
def _xavier_like(layer, next_act):
    """Initialisation with the distribution of the reference's initseq/initmod
    (core/utils/network_util.py:163-290): U(+-sqrt(3) gain sqrt(2/((n_in+n_out) k)))
    with k = kernel volume / stride volume for transposed convs, zero bias, and the
    2x2x2 block-replicated kernel for stride-2 ConvTranspose3d."""
    if isinstance(next_act, nn.ReLU):
        gain = math.sqrt(2.0)
    elif isinstance(next_act, nn.LeakyReLU):
        gain = math.sqrt(2.0 / (1.0 + next_act.negative_slope ** 2))
    else:
        gain = 1.0
    if isinstance(layer, nn.Linear):
        fan = layer.in_features + layer.out_features
    elif isinstance(layer, nn.ConvTranspose3d):
        k = layer.kernel_size[0] * layer.kernel_size[1] * layer.kernel_size[2]
        k //= layer.stride[0] * layer.stride[1] * layer.stride[2]
        fan = (layer.in_channels + layer.out_channels) * k
    else:
        return
    bound = gain * math.sqrt(2.0 / fan) * math.sqrt(3.0)
    with torch.no_grad():
        layer.weight.uniform_(-bound, bound)
        if layer.bias is not None:
            layer.bias.zero_()
        if isinstance(layer, nn.ConvTranspose3d):
            w = layer.weight
            base = w[:, :, 0::2, 0::2, 0::2].clone()
            for a in (0, 1):
                for b in (0, 1):
                    for c in (0, 1):
                        w[:, :, a::2, b::2, c::2] = base


def _init_sequence(mods):
    mods = list(mods)
    for cur, nxt in zip(mods, mods[1:] + [None]):
        _xavier_like(cur, nxt)


def _tiny_last_layer(layer, val=1e-5):
    with torch.no_grad():
        layer.weight.uniform_(-val, val)
        layer.bias.zero_()


class _Replicated(nn.Module):
    """Stands where the reference has nn.DataParallel(mlp): only there to keep
    the ``<name>.module.<...>`` parameter names."""

    def __init__(self, module):
        super().__init__()
        self.module = module


# --------------------------------------------------------------------------- parts
class _NonRigidParams(nn.Module):
    """Parameters of NonRigidMotionMLP (non_rigid_motion_mlps/mlp_offset.py:9-71)."""

    def __init__(self, pos_embed_size, condition_code_size, mlp_width, mlp_depth, skips):
        super().__init__()
        if not (pos_embed_size == 36 and condition_code_size == 69 and mlp_width == 128
                and mlp_depth == 6 and list(skips) == [4]):
            raise NotImplementedError('only the default non-rigid MLP (6x128, skip 4, PE36, cond 69) is built')
        mods = [nn.Linear(pos_embed_size + condition_code_size, mlp_width), nn.ReLU()]
        for i in range(1, mlp_depth):
            n_in = mlp_width + pos_embed_size if i in skips else mlp_width
            mods += [nn.Linear(n_in, mlp_width), nn.ReLU()]
        mods += [nn.Linear(mlp_width, 3)]
        self.block_mlps = nn.ModuleList(mods)
        _init_sequence(self.block_mlps)
        _tiny_last_layer(self.block_mlps[-1])          # mlp_offset.py:60-66

    def linears(self):
        return [m for m in self.block_mlps if isinstance(m, nn.Linear)]


class _CanonicalParams(nn.Module):
    """Parameters of CanonicalMLP default branch (canonical_mlps/mlp_rgb_sigma.py:64-99)."""

    def __init__(self, input_ch, mlp_depth, mlp_width, skips):
        super().__init__()
        if not (input_ch == 63 and mlp_depth == 8 and mlp_width == 256 and list(skips) == [4]):
            raise NotImplementedError('only the default canonical MLP (8x256, skip 4, PE63) is built')
        mods = [nn.Linear(input_ch, mlp_width), nn.ReLU()]
        for i in range(mlp_depth - 1):
            n_in = mlp_width + input_ch if i in skips else mlp_width
            mods += [nn.Linear(n_in, mlp_width), nn.ReLU()]
        self.pts_linears = nn.ModuleList(mods)
        _init_sequence(self.pts_linears)
        self.output_linear = nn.Sequential(nn.Linear(mlp_width, 4))
        _init_sequence(self.output_linear)

    def linears(self):
        return [m for m in self.pts_linears if isinstance(m, nn.Linear)] + [self.output_linear[0]]


_POINT_CONV = True       # (A/B switch of the 1x1x1 special case, profiles/tools/ab_pose.py)


class _ConvT3dK4S2P1(torch.autograd.Function):
    """Forward: the library's transposed convolution.  Backward: two plain GEMMs on the weight's NATIVE layout.

    With col[i, (co, k)] = g[co, 2 i - 1 + k] (the 4x4x4 stride-2 windows of the padded output gradient, one
    strided-view copy) the adjoint of ConvTranspose3d(4, 2, 1) is
        dx[i, ci]      = sum_(co,k) col[i, (co,k)] W[ci, (co,k)]          (D H W x 64 Cout) @ (64 Cout x Cin)
        dW[ci, (co,k)] = sum_i      x[i, ci]       col[i, (co,k)]          (Cin x D H W) @ (D H W x 64 Cout)
    -- no permuted copy of the 254 MB of decoder weights.  MIOpen has no tuned backward for these batch-1
    transposed 3-D convolutions: its solver search runs seconds of naive kernels on a fresh machine and settles
    on anything from 4 to 13 ms per training step for 20 GFLOP of work.

    A 1x1x1 input (the decoder's first layer: 1024 -> 512 channels, 33.5 M of the network's 64.4 M parameters) only
    ever meets the central 2x2x2 taps of the 4x4x4 kernel: out[co, o] = sum_ci x[ci] W[ci, co, o + 1].  That layer
    is a GEMV on those taps (16.8 MB gathered from the 134 MB tensor) instead of a pass over all 64 taps, forward and
    backward; the other 56 taps get the zero gradient they always had."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.point = _POINT_CONV and tuple(x.shape[2:]) == (1, 1, 1)
        if ctx.point:
            cin, cout = weight.shape[:2]
            wc = weight[:, :, 1:3, 1:3, 1:3].reshape(cin, cout * 8)          # (strided gather, contiguous copy)
            ctx.save_for_backward(x, wc)
            out = (x.reshape(1, cin) @ wc).reshape(1, cout, 2, 2, 2)
            return out + bias.reshape(1, cout, 1, 1, 1) if bias is not None else out
        ctx.save_for_backward(x, weight)
        if x.is_cuda and x.dtype == torch.float32 and x.shape[0] == 1:
            # GEMM on the native weight layout + the fold kernel of libhnrf (the adjoint of the backward's unfold): the same
            # two launches on every box, where MIOpen's choice for these batch-1 layers ranged from 0.15 to 1.3 ms per step
            _, cin, D, H, W = x.shape
            cout = weight.shape[1]
            col = x[0].reshape(cin, D * H * W).t() @ weight.reshape(cin, cout * 64)
            return ops.deconv_fold(col.contiguous(), None if bias is None else bias.detach().contiguous(), cout, D, H, W)
        return F.conv_transpose3d(x, weight, bias, stride=2, padding=1)

    @staticmethod
    def backward(ctx, g):
        if ctx.point:
            x, wc = ctx.saved_tensors
            cin, cout = wc.shape[0], wc.shape[1] // 8
            gm = g.reshape(1, cout * 8)
            dx = (gm @ wc.t()).reshape(1, cin, 1, 1, 1)
            dw = x.new_zeros(cin, cout, 4, 4, 4)
            dw[:, :, 1:3, 1:3, 1:3] = (x.reshape(cin, 1) @ gm).reshape(cin, cout, 2, 2, 2)
            return dx, dw, g[0].sum(dim=(1, 2, 3))
        x, weight = ctx.saved_tensors
        _, cin, D, H, W = x.shape
        cout = weight.shape[1]
        g3 = g[0]
        col = F.pad(g3, (1, 1, 1, 1, 1, 1)).unfold(1, 4, 2).unfold(2, 4, 2).unfold(3, 4, 2)   # (Cout, D, H, W, 4,4,4)
        col = col.permute(1, 2, 3, 0, 4, 5, 6).reshape(D * H * W, cout * 64)
        wm = weight.reshape(cin, cout * 64)
        xm = x[0].reshape(cin, D * H * W)
        dx = (col @ wm.t()).t().reshape(1, cin, D, H, W)
        dw = (xm @ col).reshape(cin, cout, 4, 4, 4)
        return dx, dw, g3.sum(dim=(1, 2, 3))


def conv_transpose3d_k4s2p1(x, weight, bias):
    """nn.ConvTranspose3d(kernel 4, stride 2, padding 1) of a batch-1 volume, x (1, Cin, D, H, W), with the
    GEMM backward above."""
    return _ConvT3dK4S2P1.apply(x, weight, bias)


class _PointDeconv(nn.Module):
    """ConvTranspose3d(cin, cout, 4, 2, 1) whose input is always 1x1x1: the decoder's first layer (network_util.py:30-33),
    33.5 M of the reference's 64.4 M parameters.  Such a layer only ever meets the central 2x2x2 taps of its kernel
    (out[co, o] = sum_ci x[ci] W[ci, co, o + 1]); the other 56 taps of every (ci, co) pair receive the gradient zero, so
    Adam's moments for them stay zero and so does their update: they keep their initial / loaded values for ever.
    Here the PARAMETER ``weight`` therefore holds the central taps only, (cin, cout, 2, 2, 2) -- it is what the GEMV reads
    and what autograd and the optimizer see (16.8 MB instead of 134 MB: no strided gather in the forward, no 134 MB
    zero-fill + scatter in the backward, an eighth of the optimizer's stream; bit-identical updates) -- and the full tensor
    lives on as the buffer ``weight_rest``.  ``state_dict`` / ``load_state_dict`` keep the reference's key and
    shape: 'block_conv.0.weight' (cin, cout, 4, 4, 4) with the central taps inserted / split off; GroupedAdam does the
    same for the moments in optimizer checkpoints (``hnrf_full_shape`` marks the parameter)."""

    def __init__(self, conv):
        super().__init__()
        assert isinstance(conv, nn.ConvTranspose3d) and conv.kernel_size == (4, 4, 4) and conv.stride == (2, 2, 2)
        self.in_channels, self.out_channels = conv.in_channels, conv.out_channels
        full = conv.weight.data
        self.weight = nn.Parameter(full[:, :, 1:3, 1:3, 1:3].contiguous())
        self.weight.hnrf_full_shape = tuple(full.shape)
        self.bias = nn.Parameter(conv.bias.data.clone())
        self.register_buffer('weight_rest', full.clone(), persistent=False)

    def full_weight(self):
        return expand_central_taps(self.weight.detach(), self.weight_rest)

    def forward(self, x):
        cin, cout = self.in_channels, self.out_channels
        assert tuple(x.shape) == (1, cin, 1, 1, 1)
        out = (x.reshape(1, cin) @ self.weight.reshape(cin, cout * 8)).reshape(1, cout, 2, 2, 2)
        return out + self.bias.reshape(1, cout, 1, 1, 1)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + 'weight'] = self.full_weight()
        destination[prefix + 'bias'] = self.bias if keep_vars else self.bias.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        w = state_dict.get(prefix + 'weight')
        if w is not None and tuple(w.shape) == tuple(self.weight.hnrf_full_shape):
            with torch.no_grad():
                self.weight_rest.copy_(w)
                self.weight.copy_(w[:, :, 1:3, 1:3, 1:3])
            state_dict = {k: v for k, v in state_dict.items() if k != prefix + 'weight'}
            state_dict[prefix + 'weight'] = self.weight.detach().clone()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def expand_central_taps(central, rest=None):
    """(cin, cout, 2, 2, 2) central taps -> the reference's (cin, cout, 4, 4, 4) tensor: ``rest`` (or zeros: gradients,
    optimizer moments) with the central taps replaced."""
    cin, cout = central.shape[:2]
    full = rest.clone() if rest is not None else central.new_zeros(cin, cout, 4, 4, 4)
    full[:, :, 1:3, 1:3, 1:3] = central
    return full


def full_gradient(param):
    """``param.grad`` in the reference's shape (a compact decoder parameter's gradient expanded with zeros)."""
    g = param.grad
    if g is not None and getattr(param, 'hnrf_full_shape', None) is not None and tuple(g.shape) != tuple(param.hnrf_full_shape):
        return expand_central_taps(g)
    return g


class _ConvDecoder3D(nn.Module):
    """core/utils/network_util.py:12-50."""

    def __init__(self, embedding_size, volume_size, voxel_channels):
        super().__init__()
        self.block_mlp = nn.Sequential(nn.Linear(embedding_size, 1024), nn.LeakyReLU(0.2))
        convs, cin, cout = [], 1024, 512
        for _ in range(int(math.log2(volume_size)) - 1):
            convs += [nn.ConvTranspose3d(cin, cout, 4, 2, 1), nn.LeakyReLU(0.2)]
            if cin == cout:
                cout = cin // 2
            else:
                cin = cout
        convs.append(nn.ConvTranspose3d(cin, voxel_channels, 4, 2, 1))
        self.block_conv = nn.Sequential(*convs)
        _init_sequence(self.block_mlp)
        _init_sequence(self.block_conv)
        if _POINT_CONV:
            self.block_conv[0] = _PointDeconv(self.block_conv[0])     # (after the reference's initialisation of all 64 taps)

    def forward(self, embedding):
        h = self.block_mlp(embedding).view(-1, 1024, 1, 1, 1)
        for m in self.block_conv:                                   # parameters live in the reference's modules
            h = conv_transpose3d_k4s2p1(h, m.weight, m.bias) if isinstance(m, nn.ConvTranspose3d) else m(h)
        return h


class MotionWeightVolumeDecoder(nn.Module):
    """mweight_vol_decoders/deconv_vol_decoder.py:8-33 (9 GFLOP once per frame against 40 TFLOP of per-sample
    work: stays PyTorch, with the transposed convolutions as batched GEMMs)."""

    def __init__(self, embedding_size=256, volume_size=32, total_bones=24):
        super().__init__()
        self.const_embedding = nn.Parameter(torch.randn(embedding_size), requires_grad=True)
        self.decoder = _ConvDecoder3D(embedding_size, volume_size, total_bones + 1)

    def forward(self, motion_weights_priors, **_):
        dec = self.decoder(self.const_embedding[None, ...])
        return F.softmax(dec + torch.log(motion_weights_priors), dim=1)


class BodyPoseRefiner(nn.Module):
    """pose_decoders/mlp_delta_body_pose.py:7-41."""

    def __init__(self, embedding_size=69, mlp_width=256, mlp_depth=4, total_bones=24):
        super().__init__()
        mods = [nn.Linear(embedding_size, mlp_width), nn.ReLU()]
        for _ in range(mlp_depth - 1):
            mods += [nn.Linear(mlp_width, mlp_width), nn.ReLU()]
        self.total_bones = total_bones - 1
        mods += [nn.Linear(mlp_width, 3 * self.total_bones)]
        self.block_mlps = nn.Sequential(*mods)
        _init_sequence(self.block_mlps)
        _tiny_last_layer(self.block_mlps[-1])

    def rvec(self, pose_input):
        """The MLP alone: axis-angle corrections (N * 23, 3).  One pose vector on the GPU: the single-workgroup kernels
        of libhnrf (hnrf_pose_mlp_fwd / _bwd) instead of ~30 GEMV / bias / ReLU launches per training step."""
        linears = [m for m in self.block_mlps if isinstance(m, nn.Linear)]
        if pose_input.is_cuda and pose_input.numel() == linears[0].in_features and \
                all(max(m.in_features, m.out_features) <= 256 for m in linears) and len(linears) <= 9:
            params = [p for m in linears for p in (m.weight, m.bias)]
            return _PoseMLP.apply(pose_input.reshape(-1), *params).view(-1, 3)
        return self.block_mlps(pose_input).view(-1, 3)

    def forward(self, pose_input):
        rvec = self.rvec(pose_input)
        return {'Rs': rodrigues(rvec).view(-1, self.total_bones, 3, 3),
                'rvec': rvec.view(-1, self.total_bones, 3)}


class _PoseMLP(torch.autograd.Function):
    """x (n_in,), W0, b0, W1, b1, ... -> MLP(x) on hnrf_pose_mlp_fwd; backward on hnrf_pose_mlp_bwd."""

    @staticmethod
    def forward(ctx, x, *params):
        x = x.contiguous().float()
        weights = [p.contiguous() for p in params[0::2]]
        biases = [p.contiguous() for p in params[1::2]]
        out, saved = ops.pose_mlp_fwd(x, weights, biases)
        ctx.save_for_backward(x, saved, *weights, *biases)
        ctx.n_layers = len(weights)
        return out

    @staticmethod
    def backward(ctx, g):
        x, saved = ctx.saved_tensors[:2]
        L = ctx.n_layers
        weights, biases = list(ctx.saved_tensors[2:2 + L]), list(ctx.saved_tensors[2 + L:2 + 2 * L])
        dW, db, d_x = ops.pose_mlp_bwd(g.contiguous(), x, weights, biases, saved, want_dx=ctx.needs_input_grad[0])
        grads = [None] * (2 * L)
        grads[0::2], grads[1::2] = dW, db
        return (d_x, *grads)


def rodrigues(rvec):
    """Batch Rodrigues with theta = sqrt(1e-5 + |r|^2) (network_util.py:57-83)."""
    theta = torch.sqrt(1e-5 + torch.sum(rvec ** 2, dim=1))
    r = rvec / theta[:, None]
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    oc = 1. - c
    return torch.stack((x * x + (1. - x * x) * c, x * y * oc - z * s, x * z * oc + y * s,
                        x * y * oc + z * s, y * y + (1. - y * y) * c, y * z * oc - x * s,
                        x * z * oc - y * s, y * z * oc + x * s, z * z + (1. - z * z) * c), dim=1).view(-1, 3, 3)


class _MotionBasis(torch.autograd.Function):
    """MotionBasisComputer.forward (network_util.py:125-156) and its backward on libhnrf's single-wave kernels
    (hnrf_motion_basis_fwd / _bwd) instead of ~120 tiny PyTorch launches per training step."""

    @staticmethod
    def forward(ctx, dst_Rs, dst_Ts, cnl_gtfms, rvec=None):
        dst_Rs, dst_Ts, cnl_gtfms = dst_Rs.contiguous(), dst_Ts.contiguous(), cnl_gtfms.contiguous()
        rvec = rvec.contiguous() if rvec is not None else None
        need = dst_Rs.requires_grad or dst_Ts.requires_grad or (rvec is not None and rvec.requires_grad)
        Rs, Ts, saved = ops.motion_basis_fwd(dst_Rs, dst_Ts, cnl_gtfms, want_saved=need, rvec=rvec)
        ctx.refined = rvec is not None
        if need:
            ctx.save_for_backward(dst_Rs, dst_Ts, cnl_gtfms, saved, *((rvec,) if ctx.refined else ()))
        return Rs, Ts

    @staticmethod
    def backward(ctx, g_Rs, g_Ts):
        dst_Rs, dst_Ts, cnl_gtfms, saved = ctx.saved_tensors[:4]
        rvec = ctx.saved_tensors[4] if ctx.refined else None
        out = ops.motion_basis_bwd(g_Rs.contiguous(), g_Ts.contiguous(), dst_Rs, dst_Ts, cnl_gtfms, saved, rvec=rvec)
        return out[0], out[1], None, (out[2] if ctx.refined else None)


def motion_basis(dst_Rs, dst_Ts, cnl_gtfms, rvec=None):
    """MotionBasisComputer.forward for one frame: (B,3,3),(B,3),(B,4,4) -> (B,3,3),(B,3).  ``rvec`` (B-1,3): the pose
    refinement dst_Rs[1:] <- dst_Rs[1:] Rodrigues(rvec) (network.py:677-688) applied first.  On the GPU: the fused
    kernel; the torch restatement below it serves CPU-side checks only."""
    if dst_Rs.is_cuda and dst_Rs.shape[0] == 24:
        return _MotionBasis.apply(dst_Rs, dst_Ts, cnl_gtfms, rvec)
    if rvec is not None:
        dst_Rs = torch.cat([dst_Rs[0:1], torch.matmul(dst_Rs[1:], rodrigues(rvec))], dim=0)
    return motion_basis_torch(dst_Rs, dst_Ts, cnl_gtfms)


def motion_basis_torch(dst_Rs, dst_Ts, cnl_gtfms):
    """MotionBasisComputer.forward (network_util.py:125-156) for one frame in plain torch ops:
    (B,3,3),(B,3),(B,4,4) -> (B,3,3),(B,3)."""
    B = dst_Rs.shape[0]
    G = torch.zeros(B, 4, 4, dtype=dst_Rs.dtype, device=dst_Rs.device)
    G[:, :3, :3] = dst_Rs
    G[:, :3, 3] = dst_Ts
    G[:, 3, 3] = 1.0
    chain = [G[0]]
    for i in range(1, B):
        chain.append(torch.matmul(chain[SMPL_PARENT[i]], G[i]))
    # inv_ex without the error check: torch.inverse reads the LAPACK status back to the host, a device
    # synchronisation per frame that exposes the launch latency of everything queued behind it
    f_mtx = torch.matmul(cnl_gtfms, torch.linalg.inv_ex(torch.stack(chain), check_errors=False).inverse)
    return f_mtx[:, :3, :3].contiguous(), f_mtx[:, :3, 3].contiguous()


def hann_window_weights(iter_val, multires, kick_in_iter, full_band_iter):
    """Per-band weights of the coarse-to-fine window (hannw_fourier.py:26-40),
    fp32 on the host like the reference's float32 constants."""
    kick = torch.tensor(float(kick_in_iter), dtype=torch.float32)
    t = torch.clamp(torch.tensor(float(iter_val), dtype=torch.float32) - kick, min=0.)
    N = full_band_iter - kick_in_iter
    alpha = torch.tensor(float(multires), dtype=torch.float32) if N == 0 else multires * t / N
    k = torch.arange(multires, dtype=torch.float32)
    return (1. - torch.cos(math.pi * torch.clamp(alpha - k, min=0., max=1.))) / 2.


def _versions(params):
    return tuple((p.data_ptr(), p._version) for p in params)


# --------------------------------------------------------------------------- network
def _node(root, path, default=None):
    cur = root
    for part in path.split('.'):
        if cur is None or not hasattr(cur, 'get'):
            return default
        cur = cur.get(part, None)
    return default if cur is None else cur


# every switch of the reference's Network.__init__ / CanonicalMLP / NonRigidMotionMLP constructors that selects a
# branch this build does not have (network.py:36-159, mlp_rgb_sigma.py:14-130, mlp_offset.py:9-71), with the only
# value that is built.  Module paths are compared by their last component.
_BUILT_BRANCHES = (
    ('non_rigid_motion_model', 'mlp'),
    ('canonical_mlp.view_dir', False), ('canonical_mlp.pose_color', 'wo'), ('canonical_mlp.multihead.enable', False),
    ('canonical_mlp.mlp_depth_plus', 0), ('canonical_mlp.last_linear_scale', 1), ('canonical_mlp.i_embed', 0),
    ('canonical_mlp.time_input', False), ('canonical_mlp.module', 'mlp_rgb_sigma'),
    ('non_rigid_motion_mlp.mlp_depth_plus', 0), ('non_rigid_motion_mlp.last_linear_scale', 1),
    ('non_rigid_motion_mlp.i_embed', 0), ('non_rigid_motion_mlp.time_input', False),
    ('non_rigid_motion_mlp.pose_input', True), ('non_rigid_motion_mlp.multihead.enable', False),
    ('non_rigid_motion_mlp.module', 'mlp_offset'),
    ('rgb_history.last_num', 0), ('posevec.type', 'axis_angle'), ('condition_code.type', 'global'),
    ('embedder.module', 'fourier'), ('non_rigid_embedder.module', 'hannw_fourier'),
    ('mweight_volume.module', 'deconv_vol_decoder'), ('pose_decoder.module', 'mlp_delta_body_pose'),
)


def check_config_is_built(config):
    """Raise unless ``config`` selects exactly the default-config branches of the reference network.  A checkpoint
    trained with any other setting (e.g. ``mlp_depth_plus: 2``) has extra / different tensors that a
    non-strict ``load_state_dict`` would drop silently, and the render would be wrong without an error."""
    bad = []
    for path, want in _BUILT_BRANCHES:
        have = _node(config, path, want)
        if path.endswith('.module'):
            have = str(have).split('.')[-1]
        if isinstance(want, bool):
            have = bool(have)
        if have != want:
            bad.append('%s = %r (built: %r)' % (path, have, want))
    if bad:
        raise NotImplementedError('configuration selects branches outside the hot path this build implements '
                                  '(SURVEY.md section 2.1): ' + '; '.join(bad))


class ActivationRangeError(RuntimeError):
    """A hidden activation of one of the two MLPs left the range of the split-f16 arithmetic (|x| >= 6e4; the kernels
    clamp at 65504): the frames rendered with these weights in mlp_mode 'f16x3' are not the network's output."""


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        check_config_is_built(cfg)
        cm, nr = cfg.canonical_mlp, cfg.non_rigid_motion_mlp
        self.total_bones = cfg.total_bones

        self.mweight_vol_decoder = MotionWeightVolumeDecoder(
            embedding_size=cfg.mweight_volume.embedding_size,
            volume_size=cfg.mweight_volume.volume_size,
            total_bones=cfg.total_bones)
        self.non_rigid_mlp = _Replicated(_NonRigidParams(
            pos_embed_size=6 * nr.multires, condition_code_size=nr.condition_code_size,
            mlp_width=nr.mlp_width, mlp_depth=nr.mlp_depth, skips=nr.skips))
        self.cnl_mlp = _Replicated(_CanonicalParams(
            input_ch=3 + 6 * cm.multires, mlp_depth=cm.mlp_depth, mlp_width=cm.mlp_width, skips=[4]))
        if not cfg.get('pose_decoder_off', False):
            self.pose_decoder = BodyPoseRefiner(
                embedding_size=cfg.pose_decoder.embedding_size, mlp_width=cfg.pose_decoder.mlp_width,
                mlp_depth=cfg.pose_decoder.mlp_depth, total_bones=cfg.total_bones)
        self._cnl_pack = None     # (key, packed image)
        self._nr_pack_buf = None
        # f16-range guard of inference (the training side has autograd.OperandRangeGuard): pinned copy of the two
        # images' status words, the event behind it, and the mode forced after a hit with cfg.amd.on_f16_range = 'f32'
        self._range_watch = None
        self._guard_state = None    # [canonical pack key, frames rendered with it]: the audit schedule of _guard_plan
        self._forced_mode = None
        self.f16_range_hits = self.f16_range_watched = self.f16_range_checked = 0
        self._vol_cache = None    # (key, priors, volume)
        self._workspace = None
        # set by train.Trainer when world_size > 1: dist.GradientSync whose volume_hook averages the weight-volume
        # gradient over the ranks in front of the decoder backward
        self.grad_sync = None
        # set to a list to collect (start, stop) torch.cuda.Event pairs recorded around
        # every canonical-MLP launch (bench.py roofline)
        self.mlp_event_log = None

    # reference API ---------------------------------------------------------
    def deploy_mlps_to_secondary_gpus(self):
        """No-op: every GPU runs the whole renderer (network.py:161-166)."""
        return self

    # packed-weight caches ----------------------------------------------------
    def _mlp_mode(self):
        return self._forced_mode or amd_option('mlp_mode', 'f16x3')

    # f16-range guard ---------------------------------------------------------
    def _guard_plan(self, mode, n_chunks):
        """Which ray chunks of this frame run the GUARDED kernel instances (cfg.amd.f16_range_guard; the guard costs
        3 % of the frame).  'full': all.  'off': none.  'audit' (default): all chunks of the first frame after the
        weights changed -- a checkpoint that does not fit the f16 range shows on its first frame -- then one chunk per
        frame, rotating, so that every region of the image and every pose is sampled as a sequence goes on.
        Returns (mode string for hnrf_render_frame_fwd, set of guarded chunk numbers or None = all)."""
        if mode != 'f16x3':
            return mode, None
        policy = amd_option('f16_range_guard', 'audit')
        if policy == 'full':
            return mode, None
        if policy == 'off':
            return mode + '+noguard', set()
        key = self._cnl_pack[0] if self._cnl_pack is not None else None
        if self._guard_state is None or self._guard_state[0] != key:
            self._guard_state = [key, 0]
        frame_no = self._guard_state[1]
        self._guard_state[1] += 1
        if frame_no == 0:
            return mode, None
        k = (frame_no - 1) % max(1, n_chunks)
        return mode + '+guard1:%d' % k, {k}

    def _watch_f16_range(self, cnl_packed, nr_packed, mode):
        """Called after a frame's inference kernels are queued: copies the two images' status words into a pinned slot
        -- behind the kernels in stream order, in front of the next frame's pack, which zeroes the non-rigid image's
        word -- and notes the event behind the copy.  Nothing waits: check_f16_range looks at the slots whose copy has
        landed (Network.forward does at its start, i.e. one frame late when the host keeps up with the GPU)."""
        words = [ops.status_word(cnl_packed, 'canonical', mode),
                 None if nr_packed is None else ops.status_word(nr_packed, 'nonrigid', mode)]
        if words[0] is None and words[1] is None:
            return
        if self._range_watch is None:
            import collections
            self._range_watch = {'pending': collections.deque(), 'free': []}
        w = self._range_watch
        host = w['free'].pop() if w['free'] else torch.zeros(2, dtype=torch.int32).pin_memory()
        host.zero_()
        for i, word in enumerate(words):
            if word is not None:
                host[i:i + 1].copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        w['pending'].append((host, ev))
        self.f16_range_watched += 1

    def check_f16_range(self, wait=True, upto=None):
        """Act on the status words of the watched frames whose copies have landed (``wait``: of all watched frames,
        blocking -- render loops call this once after their last frame; ``upto``: block only until the first ``upto``
        watched frames have their verdict).  ``f16_range_checked`` counts the frames whose verdict is known.  Returns
        True when one of them left the range."""
        w = self._range_watch
        if w is None:
            return False
        cnl_hit = nr_hit = 0
        while w['pending']:
            host, ev = w['pending'][0]
            if not ev.query():
                if not wait or (upto is not None and self.f16_range_checked >= upto):
                    break
                ev.synchronize()
            w['pending'].popleft()
            self.f16_range_checked += 1
            cnl_hit |= int(host[0])
            nr_hit |= int(host[1])
            w['free'].append(host)
        if not (cnl_hit or nr_hit):
            return False
        self.f16_range_hits += 1
        self._cnl_pack = None                                   # (a fresh pack clears the canonical image's word)
        policy = amd_option('on_f16_range', 'raise')
        msg = ("a hidden activation of the %s MLP reached the f16 range (|x| >= 6e4) in mlp_mode 'f16x3': the frames "
               "rendered with these weights are clamped, not the network's output; render with cfg.amd.mlp_mode = 'f32'"
               % (' and the '.join(n for n, h in (('canonical', cnl_hit), ('non-rigid', nr_hit)) if h)))
        if policy == 'ignore':
            return True
        if policy == 'f32':
            import warnings
            warnings.warn(msg + " -- switching this network to 'f32' (cfg.amd.on_f16_range = 'f32')")
            self._forced_mode = 'f32'
            return True
        raise ActivationRangeError(msg)

    def _canonical_packed(self):
        lin = self.cnl_mlp.module.linears()
        ws, bs = [l.weight for l in lin], [l.bias for l in lin]
        key = (self._mlp_mode(),) + _versions(ws + bs)
        if self._cnl_pack is None or self._cnl_pack[0] != key:
            packed = ops.canonical_pack([w.detach() for w in ws], [b.detach() for b in bs], self._mlp_mode(),
                                        out=None if self._cnl_pack is None else self._cnl_pack[1])
            self._cnl_pack = (key, packed)
        return self._cnl_pack[1]

    def _nonrigid_packed(self, cond):
        lin = self.non_rigid_mlp.module.linears()
        self._nr_pack_buf = ops.nonrigid_pack([l.weight.detach() for l in lin], [l.bias.detach() for l in lin],
                                              cond.detach().contiguous(), self._mlp_mode(), out=self._nr_pack_buf)
        return self._nr_pack_buf

    def _weight_volume(self, priors):
        """(B+1,G,G,G) softmax volume; cached across frames in eval mode."""
        params = list(self.mweight_vol_decoder.parameters())
        use_cache = (not self.training) and (not torch.is_grad_enabled()) and amd_option('cache_weight_volume', True)
        key = _versions(params)
        if use_cache and self._vol_cache is not None and self._vol_cache[0] == key \
                and self._vol_cache[1].shape == priors.shape:
            # same tensor object as last frame (a driver that keeps the priors resident): no comparison, no host
            # synchronisation; a fresh tensor is compared by value (one device round trip per frame)
            ref, ver = self._vol_cache[3]
            if ref() is priors and ver == priors._version:
                return self._vol_cache[2]
            if torch.equal(self._vol_cache[1], priors):
                # equal by value: remember THIS tensor, so that the frames that follow hit by identity (a render loop
                # uploads the subject's priors once per loop; comparing every frame blocked the host for a whole frame)
                self._vol_cache = self._vol_cache[:3] + ((weakref.ref(priors), priors._version),)
                return self._vol_cache[2]
        vol = self.mweight_vol_decoder(motion_weights_priors=priors[None])[0].contiguous()
        if use_cache:
            self._vol_cache = (key, priors.clone(), vol, (weakref.ref(priors), priors._version))
        return vol

    # forward -------------------------------------------------------------------
    def forward(self, rays, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec=None,
                near=None, far=None, iter_val=1e7, cnl_bbox_min_xyz=None, cnl_bbox_scale_xyz=None,
                bgcolor=None, t_rand=None, **kwargs):
        """network.py:647-789.  Extra keyword ``t_rand`` (N,S) injects the
        stratified-sampling uniforms (parity tests); unknown kwargs are ignored
        like the reference's **kwargs."""
        train_path = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self._range_watch is not None and self._range_watch['pending']:
            self.check_f16_range(wait=False)       # the previous frames' f16-range verdicts (may switch the mode, or raise)
        iter_val = float(iter_val)
        dev = dst_Rs.device
        f32 = lambda t: t.to(dtype=torch.float32)
        dst_Rs, dst_Ts, dst_posevec = f32(dst_Rs), f32(dst_Ts), f32(dst_posevec)
        cnl_gtfms, priors = f32(cnl_gtfms), f32(motion_weights_priors)

        # pose refinement (network.py:667-688)
        rvec = None
        if iter_val >= cfg.pose_decoder.get('kick_in_iter', 0) and not cfg.get('pose_decoder_off', False):
            rvec = self.pose_decoder.rvec(dst_posevec[None])              # (23,3); Rodrigues + correction: motion_basis

        ignore_nr = bool(cfg.ignore_non_rigid_motions)
        nr_cfg = cfg.non_rigid_motion_mlp
        cond = dst_posevec
        if iter_val < nr_cfg.kick_in_iter:
            cond = torch.zeros_like(cond) * cond                            # network.py:735-737
        motion_Rs, motion_Ts = motion_basis(dst_Rs, dst_Ts, cnl_gtfms, rvec)
        vol = self._weight_volume(priors)
        self.motion_weights_vol = vol
        if train_path and self.grad_sync is not None:
            vol = self.grad_sync.volume_hook(vol, priors)

        mode = self._mlp_mode()
        nr_packed, cnl_packed, hann_w = None, None, None
        self._nr_inputs_are_zero = False
        if not ignore_nr:
            hann_w = hann_window_weights(iter_val, nr_cfg.multires, nr_cfg.kick_in_iter, nr_cfg.full_band_iter)
            self._nr_inputs_are_zero = iter_val < nr_cfg.kick_in_iter and float(hann_w.abs().max()) == 0.0
            # through pinned memory: a pageable host-to-device copy waits for everything queued on the stream, i.e. it
            # would synchronise host and GPU once per training step / frame
            hann_w = hann_w.pin_memory().to(dev, non_blocking=True) if dev.type == 'cuda' else hann_w.to(dev)
        if not train_path:
            cnl_packed = self._canonical_packed()
            if not ignore_nr:
                nr_packed = self._nonrigid_packed(cond)

        rays_o, rays_d = rays[0], rays[1]
        rays_shape = rays_d.shape
        rays_o = f32(rays_o).reshape(-1, 3).contiguous()
        rays_d = f32(rays_d).reshape(-1, 3).contiguous()
        N = rays_o.shape[0]
        near = f32(near).reshape(-1).contiguous()
        far = f32(far).reshape(-1).contiguous()
        bbox_min = f32(cnl_bbox_min_xyz).contiguous()
        bbox_scale = f32(cnl_bbox_scale_xyz).contiguous()
        bg = f32(bgcolor).contiguous()
        S = int(cfg.N_samples)
        if cfg.perturb > 0.:
            if t_rand is None:
                t_rand = torch.rand(N, S, device=dev)                       # network.py:468
            t_rand = f32(t_rand).reshape(N, S).contiguous()
        else:
            t_rand = None
        diag = bool(amd_option('diagnostics', True))
        if N == 0:
            # a camera that does not see the subject's bbox (the reference's chunk loop, network.py:330-352, has nothing
            # to concatenate then and raises; a render loop should get its background image): the 11 keys, empty
            shp = {'rgb': (3,), 'alpha': (), 'depth': ()}
            if diag or train_path:
                shp.update(weights_on_rays=(S,), rgb_on_rays=(S, 3), cnl_xyz=(3,), cnl_rgb=(3,), cnl_weight=(),
                           xyz_on_rays=(S, 3), backward_motion_weights=(S, self.total_bones), offsets=(S, 3))
            zero = sum(p.sum() for p in self.parameters() if p.requires_grad) * 0.0 if train_path else None
            out = {k: torch.zeros((0,) + v, device=dev) + (zero if zero is not None and k in ('rgb', 'alpha', 'depth') else 0.0)
                   for k, v in shp.items()}
            return {k: v.reshape(list(rays_shape[:-1]) + list(v.shape[1:])) for k, v in out.items()}

        term_eps = float(amd_option('term_eps', 0.0))
        if not train_path and term_eps == 0.0:
            # the whole frame in one library call: chunk loop of network.py:330-352, results straight into whole-frame
            # tensors (the reference concatenates per-chunk results: one more pass over 17 KB per ray); optionally K1 of
            # the next chunk on a side stream under the MLP kernels of the current one (cfg.amd.overlap_warp)
            gmode, guarded = self._guard_plan(mode, -(-N // int(cfg.chunk)))
            out, self._workspace = ops.render_frame(
                rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w, nr_packed,
                cnl_packed, bg, S, int(cfg.chunk), gmode, diagnostics=diag,
                cull_eps=0.0 if diag else float(amd_option('cull_eps', 0.0)), workspace=self._workspace,
                overlap=bool(amd_option('overlap_warp', False)), mlp_event_log=self.mlp_event_log)
            if mode == 'f16x3' and guarded != set():
                self._watch_f16_range(cnl_packed, nr_packed, mode)
        else:
            chunks = []
            guarded = None
            if not train_path:
                _, guarded = self._guard_plan(mode, -(-N // int(cfg.chunk)))
            for ci, i in enumerate(range(0, N, int(cfg.chunk))):           # network.py:333
                sl = slice(i, min(i + int(cfg.chunk), N))
                if train_path:
                    chunks.append(self._render_rays_train(rays_o[sl], rays_d[sl], near[sl], far[sl],
                                                          None if t_rand is None else t_rand[sl], motion_Rs, motion_Ts,
                                                          vol, bbox_min, bbox_scale, hann_w, cond, bg, S, not ignore_nr,
                                                          diag))
                    continue
                chunks.append(self._render_rays(rays_o[sl], rays_d[sl], near[sl], far[sl],
                                                None if t_rand is None else t_rand[sl],
                                                motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, hann_w,
                                                nr_packed, cnl_packed, bg, S,
                                                mode if guarded is None or ci in guarded else mode + '+noguard', diag, None))
            out = {k: (torch.cat([c[k] for c in chunks], 0) if len(chunks) > 1 else chunks[0][k]) for k in chunks[0]}
            if not train_path and mode == 'f16x3' and guarded != set():
                self._watch_f16_range(cnl_packed, nr_packed, mode)
        lead = list(rays_shape[:-1])
        return {k: v.reshape(lead + list(v.shape[1:])) for k, v in out.items()}

    def _nonrigid_of_zero(self):
        """NonRigidMotionMLP.forward (mlp_offset.py:74-114) on an all-zero input row: (3,), differentiable."""
        lin = self.non_rigid_mlp.module.linears()
        # (through F.linear although W0 @ 0 = 0: W0 must receive its all-zero gradient like in the reference, so that Adam
        # creates its state and counts its steps from the first iteration -- a None gradient would restart the bias
        # correction of W0 at the kick-in iteration)
        h = F.linear(lin[0].weight.new_zeros(lin[0].in_features), lin[0].weight, lin[0].bias)
        for i, l in enumerate(lin[1:-1], start=1):
            h = torch.relu(h)
            if l.in_features != h.shape[0]:                         # the skip layer takes [h | PE36], and the PE is zero too
                h = torch.cat([h, h.new_zeros(l.in_features - h.shape[0])])
            h = F.linear(h, l.weight, l.bias)
        return F.linear(torch.relu(h), lin[-1].weight, lin[-1].bias)

    def _render_rays_train(self, rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                           hann_w, cond, bg, S, use_nonrigid, diag):
        """Differentiable chunk: autograd.RenderRays (activation-saving MLP kernels).  rgb / alpha / depth carry
        gradient (the trainer's loss reads rgb, trainer.py:121); with ``cfg.amd.diagnostics`` the other eight keys of
        the reference's return dict (network.py:776-789) are returned too, detached."""
        nr = self.non_rigid_mlp.module.linears()
        cn = self.cnl_mlp.module.linears()
        if hann_w is None:
            hann_w = torch.ones(6, device=rays_o.device)
        params = [l.weight for l in nr] + [l.bias for l in nr] + [l.weight for l in cn] + [l.bias for l in cn]
        const_offset = None
        if use_nonrigid and getattr(self, '_nr_inputs_are_zero', False):
            # before non_rigid_motion_mlp.kick_in_iter the MLP is fed zeros at every sample (condition code x 0,
            # network.py:735-737; Hann weights all 0, hannw_fourier.py:28-40): its offset is the per-frame constant MLP(0)
            # (SURVEY section 7).  Evaluated once, in fp32 with torch's autograd, instead of 786 432 times: the first
            # 10 000 (ZJU) / 100 000 (wild) iterations of a run lose the non-rigid kernels' forward, chain and weight
            # gradients (1.9 of 10 ms), and their near-zero activations never meet the f16 operand range.
            const_offset = self._nonrigid_of_zero()
            use_nonrigid = False
        res = RenderRays.apply(rays_o, rays_d, near, far, t_rand, bbox_min, bbox_scale, hann_w,
                               cond.detach().contiguous(), bg, S, use_nonrigid, diag, const_offset, motion_Rs, motion_Ts, vol,
                               *params)
        return dict(zip(RenderRays.OUTPUT_KEYS if diag else RenderRays.OUTPUT_KEYS[:3], res))

    def _render_rays(self, rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                     hann_w, nr_packed, cnl_packed, bg, S, mode, diag, dst=None):
        """network.py:474-602 for one ray chunk.  ``dst``: this chunk's row range of the whole-frame output buffers
        (full-signature path), written in place."""
        if not diag:
            term_eps = float(amd_option('term_eps', 0.0))
            need = (ops.render_term_workspace_bytes if term_eps > 0.0 else ops.render_workspace_bytes)(rays_o.shape[0], S) // 4 + 64
            if self._workspace is None or self._workspace.numel() < need or self._workspace.device != rays_o.device:
                self._workspace = torch.empty(need, device=rays_o.device)
            if term_eps > 0.0:
                return ops.render_rays_term(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min,
                                            bbox_scale, hann_w, nr_packed, cnl_packed, bg, S, mode, term_eps=term_eps,
                                            cull_eps=float(amd_option('cull_eps', 0.0)), workspace=self._workspace)
            events = None
            if self.mlp_event_log is not None:
                events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.mlp_event_log.append(events)
            return ops.render_rays(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min,
                                   bbox_scale, hann_w, nr_packed, cnl_packed, bg, S, mode,
                                   workspace=self._workspace, mlp_events=events,
                                   cull_eps=float(amd_option('cull_eps', 0.0)))
        g = (lambda k: None) if dst is None else dst.get
        z, x_skel, mask, bmw = ops.sample_warp(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol,
                                               bbox_min, bbox_scale, S, want_bmw=True,
                                               bmw_out=g('backward_motion_weights'))
        if nr_packed is not None:
            xyz, offsets = ops.nonrigid(x_skel, hann_w, nr_packed, mode, want_offsets=True,
                                        xyz_out=g('xyz_on_rays'), offsets_out=g('offsets'))
        else:
            xyz, offsets = x_skel, torch.zeros_like(x_skel)                 # network.py:276-277
            if dst is not None:
                dst['xyz_on_rays'].copy_(xyz)
                dst['offsets'].zero_()
        if self.mlp_event_log is not None:
            events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.mlp_event_log.append(events)
            events[0].record()
            raw = ops.canonical(xyz, cnl_packed, mode)
            events[1].record()
        else:
            raw = ops.canonical(xyz, cnl_packed, mode)
        out = ops.composite(raw, mask, z, rays_d, xyz, bg, diagnostics=True, out=dst)
        out.update(xyz_on_rays=xyz, backward_motion_weights=bmw, offsets=offsets)
        return out
