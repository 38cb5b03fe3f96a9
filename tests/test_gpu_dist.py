"""The multi-GPU layer on the REAL network, rehearsed on one GPU: two ranks
(two processes, gloo, both on cuda:0 -- the pool gives one card per box; with one card per rank the same code runs
over RCCL) drive the HIP path.

  (a) Trainer.backward_step on two different frames: the gradients each rank ends up with equal the mean of the two
      single-rank gradients, although the decoder's 254 MB of gradients never cross the wire
      (dist.GradientSync 'volume' mode), and both modes ('volume', 'full') agree;
  (b) after the Adam step the replicas hold the same parameters;
  (c) render.render_frames sharded over two ranks returns byte for byte the images of the serial run.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

S_TRAIN = 64


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(dev):
    from humannerf_amd.network import Network
    from humannerf_amd.seeded import default_shapes, seeded_state
    net = Network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), seed=0).items()})
    return net.to(dev)


def _train_batch(rank, dev):
    """One frame per rank (different pose, different rays, different targets), stratified uniforms injected."""
    from humannerf_amd import scene
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1250.0, ray_stride=19 + 4 * rank, pose_seed=rank)
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
            'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    b = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
    R = b['rays'].shape[1]
    rs = np.random.RandomState(50 + rank)
    b['target_rgbs'] = torch.from_numpy(rs.rand(R, 3).astype(np.float32)).to(dev)
    b['t_rand'] = torch.from_numpy(rs.rand(R, S_TRAIN).astype(np.float32)).to(dev)
    return b


def _cameras():
    from humannerf_amd import scene
    return [scene.synthetic_frame(H=96, W=96, focal_at_512=1250.0, pose_seed=i % 3, camera_only=True) for i in range(5)]


def _grads(net):
    return {n: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for n, p in net.named_parameters()}


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from humannerf_amd import render
        from humannerf_amd.config import cfg
        from humannerf_amd.train import Trainer
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        cfg.N_samples, cfg.perturb, cfg.train.lossweights.lpips = S_TRAIN, 1.0, 0.0
        cfg.amd.ddp_reduce = mode
        net = _net(dev)
        # frame-sharded render of 5 frames (before the training step changes the weights)
        cfg.N_samples, cfg.amd.diagnostics = 32, False
        imgs = render.render_frames(net, _cameras(), rank=rank, world=world, device=dev)
        cfg.N_samples, cfg.amd.diagnostics = S_TRAIN, True
        tr = Trainer(net, world_size=world)
        tr.iter = 30000                                     # non-rigid MLP active, Hann window partly open
        loss, _ = tr.backward_step(_train_batch(rank, dev))
        tr.grad_sync.finish()
        g = _grads(net)
        nbytes = tr.grad_sync.take_bytes()
        tr.optimizer_step()
        torch.cuda.synchronize()
        # replicas after the step: compare every parameter with rank 0's
        worst = 0.0
        for _, p in net.named_parameters():
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            worst = max(worst, float((ref - p.detach()).abs().max()))
        from humannerf_amd import dist as hd
        merged = hd.gather_frames({k: torch.from_numpy(v) for k, v in imgs.items()}, 5, rank, world)
        # one frame rendered by both ranks together (ray-range sharding), with the weights of before the step
        cfg.N_samples, cfg.amd.diagnostics = 32, False
        split_rgb, _ = render.render_frame_ray_sharded(_net(dev), _cameras()[1], rank, world, device=dev)
        cfg.N_samples, cfg.amd.diagnostics = S_TRAIN, True
        if rank == 0:
            q.put((g, nbytes, worst, float(loss), [m.numpy() for m in merged] + [split_rgb]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return res


@pytest.fixture(scope='module')
def serial():
    """Single-process references: mean of the two single-rank gradients, serial render of the 5 frames."""
    from humannerf_amd import render
    from humannerf_amd.config import cfg
    from humannerf_amd.train import Trainer
    dev = torch.device('cuda:0')
    old = (cfg.N_samples, cfg.perturb, cfg.train.lossweights.lpips, cfg.amd.diagnostics)
    cfg.N_samples, cfg.perturb, cfg.train.lossweights.lpips = S_TRAIN, 1.0, 0.0
    try:
        cfg.N_samples, cfg.amd.diagnostics = 32, False
        imgs = render.render_frames(_net(dev), _cameras(), device=dev)
        cfg.N_samples, cfg.amd.diagnostics = S_TRAIN, True
        acc = None
        for rank in range(2):
            net = _net(dev)
            tr = Trainer(net, world_size=1)
            tr.iter = 30000
            tr.backward_step(_train_batch(rank, dev))
            g = _grads(net)
            acc = g if acc is None else {n: (None if g[n] is None else (acc[n] + g[n]) / 2) for n in g}
            del tr
    finally:
        cfg.N_samples, cfg.perturb, cfg.train.lossweights.lpips, cfg.amd.diagnostics = old
    return acc, [imgs[i] for i in range(5)]


@pytest.mark.parametrize('mode', ['volume', 'full'])
def test_world2_gradients_replicas_and_sharded_render(mode, serial):
    want, want_imgs = serial
    grads, nbytes, worst, loss, imgs = _run(mode)
    assert np.isfinite(loss)
    assert set(grads) == set(want)
    for n in want:
        assert (grads[n] is None) == (want[n] is None), n
        if want[n] is None:
            continue
        scale = float(np.abs(want[n]).max())
        err = float(np.abs(grads[n] - want[n]).max())
        # same kernels on the same inputs; what differs is the fp32 order in which the two frames are summed
        # (before instead of after the decoder backward) and the atomics of the volume-gradient kernel
        assert err <= 2e-5 * scale + 1e-12, (n, err, scale)
    dec = 4 * (63589145 - 1024 * 512 * 56)                 # the decoder's trainable parameters (network._PointDeconv)
    if mode == 'volume':
        assert nbytes < 8 * 1024 * 1024, nbytes            # 3.3 MB volume gradient + 3.3 MB bucket
    else:
        assert nbytes > dec
    assert worst <= 1e-7, worst                            # replicas stay together after the Adam step
    assert len(imgs) == 6
    assert np.array_equal(imgs.pop(), want_imgs[1])        # the ray-sharded frame: byte for byte the serial image
    for a, b in zip(imgs, want_imgs):
        assert a.dtype == np.uint8 and a.shape == (96, 96, 3)
        assert np.array_equal(a, b)
    assert imgs[0].std() > 1.0                             # not an empty image


def _rccl_worker(port, q):
    """One rank, backend 'nccl' (= RCCL): the collectives of GradientSync run for real (over one rank they are
    identities), inside autograd's backward thread and on the training stream, as they do with one GPU per rank."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from humannerf_amd.config import cfg
        from humannerf_amd.train import Trainer
        cfg.N_samples, cfg.perturb, cfg.train.lossweights.lpips = S_TRAIN, 1.0, 0.0
        out = {}
        for mode in ('none', 'volume', 'full'):
            cfg.amd.ddp_reduce = mode if mode != 'none' else 'volume'
            cfg.amd.ddp_single_rank_collectives = mode != 'none'
            net = _net(dev)
            tr = Trainer(net, world_size=1)
            assert tr.grad_sync.active == (mode != 'none')
            tr.iter = 30000
            losses, g = [], None
            for _ in range(2):                              # second step: persistent buckets, deferred flags polled
                loss, _ = tr.backward_step(_train_batch(0, dev))
                g = _grads(net) if g is None else g          # (compared: the first step's, from identical parameters)
                tr.optimizer_step()
                losses.append(float(loss))
            tr.grad_sync.finish()
            torch.cuda.synchronize()
            out[mode] = (g, losses, tr.grad_sync.take_bytes())
        dist.barrier()
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_collectives_run_on_rccl():
    """The same training step over backend 'nccl' (RCCL) with one rank: results equal the collective-free run, and the
    byte counters show that the collectives were issued."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    out = q.get()
    p.join(300)
    assert p.exitcode == 0
    ref, ref_losses, ref_bytes = out['none']
    assert ref_bytes == 0
    for mode in ('volume', 'full'):
        g, losses, nbytes = out[mode]
        assert nbytes > 3 * 1024 * 1024, (mode, nbytes)
        assert np.allclose(losses, ref_losses, rtol=1e-3)
        for n in ref:
            assert (g[n] is None) == (ref[n] is None), n
            if ref[n] is not None:
                scale = float(np.abs(ref[n]).max())
                assert float(np.abs(g[n] - ref[n]).max()) <= 2e-5 * scale + 1e-12, (mode, n)
