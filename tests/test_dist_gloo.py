"""world_size-2 gloo tests of the multi-GPU layer (runs on CPU).

The renderer itself needs the GPU (tests/test_gpu_dist.py does the same with the real Network on the box); here the
GradientSync logic is exercised on the real weight-volume decoder (a 16^3 instance) with a stand-in for the
per-sample path: any differentiable function of (volume, small parameters) will do, because what is being
checked is that averaging the volume gradient in front of the decoder backward + one flat bucket for the rest
gives exactly the mean of the per-rank gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    """mweight_vol_decoder (real, 16^3) + two small parameter sets named like the network's."""

    def __init__(self, seed=1):
        super().__init__()
        from humannerf_amd.network import MotionWeightVolumeDecoder
        torch.manual_seed(seed)
        self.mweight_vol_decoder = MotionWeightVolumeDecoder(embedding_size=32, volume_size=16, total_bones=24)
        self.cnl_mlp = torch.nn.Linear(6, 4)
        self.pose_decoder = torch.nn.Linear(5, 3)          # gets NO gradient in `loss` when use_pose is False
        self.grad_sync = None

    def loss(self, priors, seed, use_pose=True):
        g = torch.Generator().manual_seed(seed)
        vol = self.mweight_vol_decoder(motion_weights_priors=priors[None])[0]
        if self.grad_sync is not None:
            vol = self.grad_sync.volume_hook(vol, priors)
        a = torch.randn(vol.shape, generator=g)
        x = torch.randn(9, 6, generator=g)
        out = (vol * a).sum() * 0.01 + self.cnl_mlp(x).pow(2).sum() * (vol[:24] * a[:24]).mean()
        if use_pose:
            out = out + self.pose_decoder(torch.randn(2, 5, generator=g)).sum()
        return out


def _priors(seed=0):
    g = torch.Generator().manual_seed(100 + seed)
    p = torch.rand(25, 16, 16, 16, generator=g) + 0.05
    return p / p.sum(0, keepdim=True)


def _worker(rank, world, port, q, mode, use_pose, priors_differ):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from humannerf_amd import dist as hd
    try:
        # frame sharding + ordered gather: stand-in renderer = deterministic function of the frame index
        n_frames = 7
        mine = hd.frame_shard(n_frames, rank, world)
        local = {i: torch.full((4, 3), float(i)) * (i + 1) for i in mine}
        frames = hd.gather_frames(local, n_frames, rank, world)

        net = _Toy()
        sync = hd.GradientSync(net, world, mode=mode)
        net.grad_sync = sync
        err = None
        try:
            net.loss(_priors(rank if priors_differ else 0), seed=10 + rank, use_pose=use_pose).backward()
            sync.reduce()
            sync.finish()
        except RuntimeError as e:
            err = str(e)
        grads = {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in net.named_parameters()}   # by value: the sender may exit first
        nbytes = sync.take_bytes()
        if rank == 0:
            q.put(([f.tolist() for f in frames] if frames is not None else None, grads, nbytes, err))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(mode, use_pose=True, priors_differ=False, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, use_pose, priors_differ)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _serial_mean(use_pose=True, world=2):
    torch.set_num_threads(2)
    acc = None
    for rank in range(world):
        net = _Toy()
        net.loss(_priors(0), seed=10 + rank, use_pose=use_pose).backward()
        g = {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in net.named_parameters()}
        acc = g if acc is None else {n: (None if g[n] is None else acc[n] + g[n]) for n in g}
    return {n: (None if v is None else v / world) for n, v in acc.items()}


@pytest.mark.parametrize('mode', ['volume', 'full'])
def test_gradient_sync_equals_mean_of_rank_gradients(mode):
    frames, grads, nbytes, err = _run(mode)
    assert err is None, err
    assert [f[0][0] for f in frames] == [float(i) * (i + 1) for i in range(7)]     # serial frame order
    want = _serial_mean()
    assert set(grads) == set(want)
    for n in want:
        scale = float(abs(want[n]).max()) + 1e-12
        assert float(abs(grads[n] - want[n]).max()) <= 2e-6 * scale, n
    n_dec = sum(p.numel() for n, p in _Toy().named_parameters() if 'mweight_vol_decoder' in n)
    if mode == 'volume':
        # the decoder's own gradients never travel: volume (25*16^3) + small bucket only
        assert nbytes < 4 * (25 * 16 ** 3 + 200) and nbytes < 4 * n_dec / 10
    else:
        assert nbytes > 4 * n_dec


def test_gradient_sync_at_world_4():
    """The same with four ranks (what BASELINE config 3 runs): mean of four rank gradients, frames dealt 4 ways."""
    frames, grads, nbytes, err = _run('volume', world=4)
    assert err is None, err
    assert [f[0][0] for f in frames] == [float(i) * (i + 1) for i in range(7)]
    want = _serial_mean(world=4)
    for n in want:
        scale = float(abs(want[n]).max()) + 1e-12
        assert float(abs(grads[n] - want[n]).max()) <= 4e-6 * scale, n


def test_parameters_without_gradient_stay_none():
    """a parameter no rank produced a gradient for must not be stepped (grad stays None)."""
    _, grads, _, err = _run('volume', use_pose=False)
    assert err is None, err
    assert grads['pose_decoder.weight'] is None and grads['pose_decoder.bias'] is None
    want = _serial_mean(use_pose=False)
    assert float(abs(grads['cnl_mlp.weight'] - want['cnl_mlp.weight']).max()) <= 1e-6


def test_volume_mode_detects_rank_dependent_priors():
    """The volume trick needs replicated decoder activations; different priors per rank must raise."""
    _, _, _, err = _run('volume', priors_differ=True)
    assert err is not None and 'priors differ' in err


def _worker_seeds(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from humannerf_amd import dist as hd
    try:
        net = _Toy(seed=100 + rank)                       # what train_subject does without a checkpoint: seed + rank
        before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
        sync = hd.GradientSync(net, world, mode='volume')
        net.grad_sync = sync
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        net.loss(_priors(0), seed=10 + rank).backward()
        sync.reduce()
        opt.step()
        sync.finish()
        after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        q.put((rank, before.numpy().copy(), after.numpy().copy(),
               torch.cat([p.detach().reshape(-1) for p in _Toy(seed=100).parameters()]).numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicas_start_equal_whatever_the_ranks_initialised():
    """nothing synchronised the initial parameters -- a run that does not resume builds each rank's network
    from its own seed, and the volume mode (decoder gradients computed locally from the averaged volume gradient)
    silently assumes bit-identical decoder weights.  GradientSync now broadcasts rank 0's state on construction:
    ranks built with DIFFERENT seeds hold rank 0's parameters afterwards and are still equal after an Adam step."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker_seeds, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get() for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, b0, a0, ref0), (_, b1, a1, _) = res
    assert not (b0 == b1).all()                           # the ranks really started apart
    assert (a0 == a1).all()                               # ... and are replicas after construction + one step
    assert (b0 == ref0).all() and not (a0 == b0).all()    # rank 0's initial state won, and the step moved it


def test_frame_shard_covers_everything():
    from humannerf_amd.dist import frame_shard
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in frame_shard(100, r, world))
        assert got == list(range(100))
        sizes = [len(frame_shard(100, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
