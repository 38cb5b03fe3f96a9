"""world_size-2 gloo tests of the multi-GPU layer (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from humannerf_amd import dist as hd
    # frame sharding + ordered gather: stand-in renderer = deterministic function of the frame index
    n_frames = 7
    mine = hd.frame_shard(n_frames, rank, world)
    local = {i: torch.full((4, 3), float(i)) * (i + 1) for i in mine}
    frames = hd.gather_frames(local, n_frames, rank, world)
    # gradient all-reduce == mean of the per-rank grads, both buckets
    torch.manual_seed(0)
    params = [('mweight_vol_decoder.w', torch.nn.Parameter(torch.zeros(5, 3))),
              ('cnl_mlp.module.w', torch.nn.Parameter(torch.zeros(7))),
              ('pose_decoder.b', torch.nn.Parameter(torch.zeros(2)))]
    for k, (_, p) in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (k + 1))
    hd.allreduce_gradients(params, world)
    if rank == 0:
        q.put(([f.tolist() for f in frames], [p.grad.tolist() for _, p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_shard_and_grad_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames, grads = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [f[0][0] for f in frames] == [float(i) * (i + 1) for i in range(7)]     # serial order
    mean = (1 + 2) / 2.0
    assert grads[0][0][0] == mean * 1 and grads[1][0] == mean * 2 and grads[2][0] == mean * 3


def test_frame_shard_covers_everything():
    from humannerf_amd.dist import frame_shard
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in frame_shard(100, r, world))
        assert got == list(range(100))
        sizes = [len(frame_shard(100, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
