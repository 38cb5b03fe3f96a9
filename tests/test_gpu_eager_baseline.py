"""The "reference PyTorch renderer on one MI355X" comparator (BASELINE.md
section 4 item 3): the op-faithful restatement of the reference's torch-op sequence (same 32768-ray /
300000-sample chunking), executed eagerly by PyTorch-ROCm on the GPU, timed against the HIP path on
the same 512x512x128 frame.  It is a restatement, not the reference source (which cannot travel to the
GPU box).  north_star target: >= 10x."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_path_vs_eager_pytorch_rocm(seeded_params):
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    from oracle import oracle
    dev = torch.device('cuda:0')
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
    R = fr['rays'].shape[1]
    state_gpu = {k: torch.from_numpy(v).to(dev) for k, v in seeded_params.items()}

    def eager():
        with torch.no_grad():
            return oracle.render(state_gpu, fr, iter_val=1e7, N_samples=128, device=dev, use_grid_sample=True)
    eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = eager()
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0

    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(dev).eval()
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
    cfg.perturb, cfg.N_samples = 0., 128
    res, errs = {}, {}
    try:
        for mode in ('f16x3', 'f32'):
            cfg.amd.mlp_mode = mode
            cfg.amd.diagnostics = True            # like for like: all 11 outputs, as the eager restatement produces
            with torch.no_grad():
                net(**data, iter_val=1e7)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    out = net(**data, iter_val=1e7)
                torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / 3
            err = float((out['rgb'] - ref['rgb']).abs().max())
            errs[mode] = err
            # two fp32 evaluations against each other on all 262 144 rays: each is held to 2e-5 against the reference
            # (tests/test_gpu_parity.py), so 4e-5 between them; measured 2.6e-5 (f16x3) on the driver's bench frame
            assert err <= 4e-5, (mode, err)
    finally:
        cfg.amd.mlp_mode, cfg.amd.diagnostics, cfg.perturb = 'f16x3', True, 1.0
    print('\neager PyTorch-ROCm restatement: %.1f ms/frame = %.0f rays/s' % (t_eager * 1e3, R / t_eager))
    for mode, t in res.items():
        print('HIP path %-6s: %.1f ms/frame = %.0f rays/s  (%.1fx eager), max |d rgb| vs eager %.2e'
              % (mode, t * 1e3, R / t, t_eager / t, errs[mode]))
    # measured round 1: eager ~0.7-1.2 s/frame; HIP f16x3 ~0.13 s, f32 ~0.35 s
    assert t_eager / res['f16x3'] >= 4.0
    assert t_eager / res['f32'] >= 1.5


def test_training_step_vs_eager_pytorch_rocm(seeded_params):
    """Second metric of BASELINE.json (train iters/s): one optimisation step on 6 144 rays x 128 samples
    (default.yaml:352-357: 6 patches of 32x32) -- eager torch.autograd through the op-faithful restatement
    vs the HIP training path (autograd.RenderRays), same Adam step on the same parameters."""
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    from humannerf_amd.train import Trainer
    from oracle import oracle
    dev = torch.device('cuda:0')
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0)
    idx = (np.arange(6144) * 37) % (512 * 512)
    sub = dict(fr)
    sub['rays'], sub['near'], sub['far'] = fr['rays'][:, idx], fr['near'][idx], fr['far'][idx]
    target = torch.rand(6144, 3, device=dev)

    state = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in seeded_params.items()}
    opt = torch.optim.Adam(list(state.values()), lr=5e-4)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        out = oracle.render(state, sub, iter_val=1e7, N_samples=128, device=dev, use_grid_sample=True)
        loss = 0.2 * torch.mean((out['rgb'] - target) ** 2)
        loss.backward()
        opt.step()
    for _ in range(2):
        eager_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eager_step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 3

    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params.items()})
    net = net.to(dev).train()
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    batch = {k: torch.from_numpy(np.ascontiguousarray(sub[k])).to(dev) for k in keys}
    batch['target_rgbs'] = target
    cfg.perturb, cfg.N_samples = 1.0, 128
    tr = Trainer(net)
    for _ in range(3):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t_hip = (time.perf_counter() - t0) / 5
    print('\ntraining step: eager PyTorch-ROCm %.1f ms (%.1f it/s), HIP path %.1f ms (%.1f it/s): %.1fx'
          % (t_eager * 1e3, 1 / t_eager, t_hip * 1e3, 1 / t_hip, t_eager / t_hip))
    # measured round 1: eager ~150 ms, HIP ~36 ms
    assert t_eager / t_hip >= 2.0
