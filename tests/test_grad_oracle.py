"""Training path, CPU side: torch.autograd through the oracle reproduces the gradients the
REFERENCE produced for the same scalar loss (tests/golden/grad_s64.npz, from
oracle/make_golden.py).  This pins the oracle as the gradient checker used on the GPU box."""
import json
import os

import numpy as np
import torch

from oracle import oracle


def reference_loss(out, lw):
    lw = torch.as_tensor(lw)
    return (out['rgb'] * lw[:, :3].to(out['rgb'])).sum() + (out['alpha'] * lw[:, 3].to(out['rgb'])).sum() \
        + 0.1 * (out['depth'] * lw[:, 4].to(out['rgb'])).sum()


def grad_frame(meta):
    from humannerf_amd import scene
    return scene.synthetic_frame(H=512, W=512, focal_at_512=meta['focal_at_512'], ray_stride=meta['ray_stride'])


def oracle_gradients(seeded_params, meta, g, dtype):
    """Gradients of the fixture's scalar loss by torch.autograd through the oracle in ``dtype``: name -> ndarray, loss."""
    fr = grad_frame(meta)
    state = {k: torch.from_numpy(v).to(dtype).clone().requires_grad_(True) for k, v in seeded_params.items()}
    out = oracle.render(state, fr, iter_val=meta['iter_val'], N_samples=meta['N_samples'], dtype=dtype)
    loss = reference_loss(out, g['loss_weights'])
    loss.backward()
    return {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in state.items()}, float(loss)


def compare_exact(grads, exact, rel_norm, cos_min, loose=None):
    """Every tensor of ``grads`` against the fp64 evaluation ``exact``: relative norm error and cosine, whole tensors.
    ``loose``: (name prefix, rel_norm, cos_min) for the tensors with their own bound."""
    worst_norm, worst_cos = (0.0, None), (1.0, None)
    tight = (rel_norm, cos_min)
    for name, ref in exact.items():
        rel_norm, cos_min = (loose[1], loose[2]) if loose is not None and name.startswith(loose[0]) else tight
        a, r = grads[name].astype(np.float64).reshape(-1), ref.astype(np.float64).reshape(-1)
        na, nr = np.linalg.norm(a), np.linalg.norm(r)
        assert nr > 0, name
        cos = float(a @ r / (na * nr))
        worst_norm, worst_cos = max(worst_norm, (abs(na - nr) / nr, name)), min(worst_cos, (cos, name))
        assert abs(na - nr) <= rel_norm * nr and cos >= cos_min, (name, na, nr, cos)
    return {'worst_rel_norm_err': worst_norm, 'worst_cosine': worst_cos}


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
    assert grad_frame(meta)['rays'].shape[1] == meta['n_rays']
    grads, loss = oracle_gradients(seeded_params, meta, g, torch.float32)
    assert abs(loss - meta['loss']) <= 1e-4 * max(1.0, abs(meta['loss']))
    compare_grads(grads, g)


def test_reference_gradient_noise_floor(seeded_params, golden_dir):
    """How far the reference's own fp32 gradients are from an fp64 evaluation of the same formulas: up to 4.4e-3 in
    norm (pose decoder, non-rigid MLP: they sit behind the 2^9 positional-encoding band), cosine >= 0.99999.  This is
    the floor any bound against the reference-gradient fixture has to respect; the GPU kernels are held to the fp64
    evaluation much more tightly (tests/test_gpu_grad.py)."""
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        meta = json.load(f)['grad_s64']
    g = np.load(os.path.join(golden_dir, 'grad_s64.npz'))
    grads, _ = oracle_gradients(seeded_params, meta, g, torch.float64)
    stats = compare_grads(grads, g, rel_norm=6e-3, cos_min=0.99998)
    print('fp64 oracle vs reference fp32 gradients', stats)
    assert stats['worst_rel_norm_err'][0] > 2e-3          # i.e. 2e-3 would NOT be a meaningful bound vs the reference
