"""Training path, CPU side: torch.autograd through the oracle reproduces the gradients the
REFERENCE produced for the same scalar loss (tests/golden/grad_s64.npz, from
oracle/make_golden.py).  This pins the oracle as the gradient checker used on the GPU box."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle


def reference_loss(out, lw):
    lw = torch.as_tensor(lw)
    return (out['rgb'] * lw[:, :3].to(out['rgb'])).sum() + (out['alpha'] * lw[:, 3].to(out['rgb'])).sum() \
        + 0.1 * (out['depth'] * lw[:, 4].to(out['rgb'])).sum()


def grad_frame(meta):
    from humannerf_amd import scene
    return scene.synthetic_frame(H=512, W=512, focal_at_512=meta['focal_at_512'], ray_stride=meta['ray_stride'])


def compare_grads(grads, g, rel_norm=2e-3, cos_min=0.9995):
    """grads: name -> ndarray.  g: the golden npz."""
    checked, worst_norm, worst_cos = 0, (0.0, None), (1.0, None)
    for key in g.files:
        if not key.startswith('norm/'):
            continue
        name = key[5:]
        ref_norm = float(g[key])
        got = grads[name]
        gn = float(np.linalg.norm(got.astype(np.float64)))
        assert abs(gn - ref_norm) <= rel_norm * max(ref_norm, 1e-8), (name, gn, ref_norm)
        if ref_norm > 0:
            worst_norm = max(worst_norm, (abs(gn - ref_norm) / ref_norm, name))
        ref = g['grad/' + name] if 'grad/' + name in g.files else None
        a = got.reshape(-1)
        if ref is None and 'head/' + name in g.files:
            ref, a = g['head/' + name], a[:4096]
        if ref is not None and np.linalg.norm(ref) > 0:
            r = ref.reshape(-1).astype(np.float64)
            cos = float(a.astype(np.float64) @ r / (np.linalg.norm(a) * np.linalg.norm(r) + 1e-300))
            assert cos >= cos_min, (name, cos)
            worst_cos = min(worst_cos, (cos, name))
        checked += 1
    assert checked == 55
    return {'worst_rel_norm_err': worst_norm, 'worst_cosine': worst_cos}


def test_oracle_autograd_matches_reference(seeded_params, golden_dir):
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)['grad_s64']
    g = np.load(os.path.join(golden_dir, 'grad_s64.npz'))
    fr = grad_frame(meta)
    assert fr['rays'].shape[1] == meta['n_rays']
    state = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in seeded_params.items()}
    out = oracle.render(state, fr, iter_val=meta['iter_val'], N_samples=meta['N_samples'])
    loss = reference_loss(out, g['loss_weights'])
    assert abs(float(loss) - meta['loss']) <= 1e-4 * max(1.0, abs(meta['loss']))
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in state.items()}
    compare_grads(grads, g)
