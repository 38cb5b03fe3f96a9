"""Seeded synthetic parameters: the random-init weights of the default architecture that the benchmark, the
smoke check, the golden-vector generator and the tests all use (a data recipe, no arithmetic of the path).

The reference ships no checkpoint we can fetch, and a 258 MB state_dict cannot
be committed, so both sides (the imported reference in ``make_golden.py`` and
this repo's ``Network`` in the tests) fill their parameters from the same
``numpy.random.RandomState`` stream: tensors visited in sorted-name order,
Xavier-uniform-like scale from the tensor's shape, small non-zero biases, then
the per-tensor multipliers below so that densities, offsets and pose
corrections are non-degenerate (fresh reference init gives sigma ~ 0 and
offsets ~ 1e-5).
"""
import math

import numpy as np

# name-substring -> (weight multiplier, bias override or None)
_TWEAKS = {
    # keep the decoded volume close to (prior x smooth perturbation)
    'mweight_vol_decoder.decoder.block_conv.8.weight': 0.6,
    # non-rigid offsets of a few centimetres
    'non_rigid_mlp.module.block_mlps.12.weight': 0.02,
    # pose correction of a few degrees per joint
    'pose_decoder.block_mlps.8.weight': 0.05,
    # larger logits so that alpha / colours spread over (0,1)
    'cnl_mlp.module.output_linear.0.weight': 3.0,
}
SIGMA_BIAS = 15.0  # added to cnl_mlp output_linear bias[3]
SIGMA_GAIN = 12.0  # extra multiplier of the sigma row of output_linear.weight


def _bound(name, shape):
    if len(shape) == 1:
        return None
    if len(shape) == 2:                       # nn.Linear (out, in)
        return math.sqrt(2.0) * math.sqrt(6.0 / (shape[0] + shape[1]))
    if len(shape) == 5:                       # ConvTranspose3d (in, out, 4,4,4), stride 2
        ksize = (shape[2] * shape[3] * shape[4]) // 8
        return math.sqrt(2.0) * math.sqrt(6.0 / ((shape[0] + shape[1]) * ksize))
    raise ValueError((name, shape))


def seeded_state(shapes, seed=0):
    """shapes: mapping name -> tuple.  Returns name -> float32 ndarray."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith('const_embedding'):
            v = rs.randn(*shape)
        elif len(shape) == 1:
            v = rs.uniform(-0.05, 0.05, size=shape)
        else:
            b = _bound(name, shape)
            v = rs.uniform(-b, b, size=shape)
        for key, mult in _TWEAKS.items():
            if key in name:
                v = v * mult
        if name == 'cnl_mlp.module.output_linear.0.bias':
            v[3] += SIGMA_BIAS
        if name == 'cnl_mlp.module.output_linear.0.weight':
            v[3] *= SIGMA_GAIN
        out[name] = v.astype(np.float32)
    return out


def with_density(state, bias_delta=0.0, gain=1.0):
    """Variant of a seeded state with a denser (or thinner) medium: the sigma row of the canonical head scaled by
    ``gain`` and its bias moved by ``bias_delta``, in float32 on the float32 base tensors (so that the fixture
    generator and the tests derive bit-identical weights).  Every other tensor is shared with ``state``."""
    out = dict(state)
    w = state['cnl_mlp.module.output_linear.0.weight'].copy()
    b = state['cnl_mlp.module.output_linear.0.bias'].copy()
    w[3] = w[3] * np.float32(gain)
    b[3] = b[3] + np.float32(bias_delta)
    out['cnl_mlp.module.output_linear.0.weight'], out['cnl_mlp.module.output_linear.0.bias'] = w, b
    return out


def default_shapes(volume_size=32, total_bones=24):
    """Parameter names/shapes of the reference Network in its default config
    (SURVEY.md section 5 key-name contract; Appendix A.4 census = 64 417 381 at 32^3)."""
    s = {'mweight_vol_decoder.const_embedding': (256,),
         'mweight_vol_decoder.decoder.block_mlp.0.weight': (1024, 256),
         'mweight_vol_decoder.decoder.block_mlp.0.bias': (1024,)}
    cin, cout, i = 1024, 512, 0
    for _ in range(int(math.log2(volume_size)) - 1):
        s[f'mweight_vol_decoder.decoder.block_conv.{i}.weight'] = (cin, cout, 4, 4, 4)
        s[f'mweight_vol_decoder.decoder.block_conv.{i}.bias'] = (cout,)
        if cin == cout:
            cout = cin // 2
        else:
            cin = cout
        i += 2
    s[f'mweight_vol_decoder.decoder.block_conv.{i}.weight'] = (cin, total_bones + 1, 4, 4, 4)
    s[f'mweight_vol_decoder.decoder.block_conv.{i}.bias'] = (total_bones + 1,)
    nr = [(128, 105), (128, 128), (128, 128), (128, 128), (128, 164), (128, 128), (3, 128)]
    for n, sh in enumerate(nr):
        s[f'non_rigid_mlp.module.block_mlps.{2 * n}.weight'] = sh
        s[f'non_rigid_mlp.module.block_mlps.{2 * n}.bias'] = (sh[0],)
    cn = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 2
    for n, sh in enumerate(cn):
        s[f'cnl_mlp.module.pts_linears.{2 * n}.weight'] = sh
        s[f'cnl_mlp.module.pts_linears.{2 * n}.bias'] = (sh[0],)
    s['cnl_mlp.module.output_linear.0.weight'] = (4, 256)
    s['cnl_mlp.module.output_linear.0.bias'] = (4,)
    pd = [(256, 69), (256, 256), (256, 256), (256, 256), (3 * (total_bones - 1), 256)]
    for n, sh in enumerate(pd):
        s[f'pose_decoder.block_mlps.{2 * n}.weight'] = sh
        s[f'pose_decoder.block_mlps.{2 * n}.bias'] = (sh[0],)
    return s
