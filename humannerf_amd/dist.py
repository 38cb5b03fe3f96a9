"""Multi-GPU layer of the path: one process per GPU, whole renderer replicated.

* Rendering shards FRAMES (rays of independent frames) across ranks -- no
  data-path collective; rank 0 only gathers finished images on the host
  (BASELINE.json config 5).
* Training (next round) all-reduces gradients once per step over RCCL
  (``backend='nccl'`` is RCCL on ROCm; ``gloo`` on CPU in the tests) in two flat
  buckets: the 254 MB ConvTranspose3d decoder and the 3.3 MB of MLPs.

The reference has no distributed code at all (its only mechanism is
nn.DataParallel over the two MLPs, network.py:68-72,115-119).
"""
import torch
import torch.distributed as dist


def frame_shard(n_frames, rank, world):
    """Frames rendered by ``rank``: round-robin so consecutive frames of a
    sequence land on different GPUs (movement / freeview renders)."""
    return list(range(rank, n_frames, world))


def gather_frames(local, n_frames, rank, world, dst=0):
    """Collect per-rank ``{frame_idx: tensor}`` dicts on ``dst`` in frame order
    (host-side gather of finished images; not on the data path)."""
    if world == 1:
        return [local[i] for i in range(n_frames)]
    objs = [None] * world if rank == dst else None
    dist.gather_object({k: v.cpu() for k, v in local.items()}, objs, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for d in objs:
        merged.update(d)
    return [merged[i] for i in range(n_frames)]


def gradient_buckets(named_params):
    """Two buckets: the big decoder (ready last in backward, it is the first op of
    forward) and everything else (ready first; small enough for one shot)."""
    big, small = [], []
    for name, p in named_params:
        if p.requires_grad:
            (big if 'mweight_vol_decoder' in name else small).append(p)
    return [b for b in (small, big) if b]


def allreduce_gradients(named_params, world=None):
    """Mean-all-reduce of .grad over all ranks, one flat collective per bucket."""
    world = world or (dist.get_world_size() if dist.is_initialized() else 1)
    if world == 1:
        return
    for bucket in gradient_buckets(list(named_params)):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for p, g in zip(bucket, grads):
            n = g.numel()
            p.grad = flat[off:off + n].view_as(p).clone()
            off += n
