"""Multi-GPU layer of the path: one process per GPU, whole renderer replicated.

* Rendering shards FRAMES (rays of independent frames) across ranks -- no
  data-path collective; rank 0 only gathers finished images on the host
  (BASELINE.json config 5).
* Training: every rank renders its own frame, gradients are mean-all-reduced over
  RCCL (``backend='nccl'`` is RCCL on ROCm; ``gloo`` in the tests) -- see
  ``GradientSync``: 6.6 MB per step instead of the 257.7 MB of a plain
  all-reduce over every parameter.

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
    dist.gather_object({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in local.items()}, objs, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for d in objs:
        merged.update(d)
    return [merged[i] for i in range(n_frames)]


class _MeanAllReduceGrad(torch.autograd.Function):
    """Identity in forward; the incoming gradient is mean-all-reduced over the ranks in backward."""

    @staticmethod
    def forward(ctx, x, sync):
        ctx.sync = sync
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # a copy: autograd hands the same tensor to every other consumer of this gradient (hooks, retain_grad), a
        # backward must not write into it (3.3 MB)
        g = g.clone(memory_format=torch.contiguous_format)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.sync.group)
        return g.div_(ctx.sync.world), None


class GradientSync:
    """Gradient averaging of one training step, laid out for what the step actually produces.

    98.7 % of the parameters (63.6 M of 64.4 M, 254 MB of gradient) belong to the motion-weight-volume
    decoder, whose input is a learned constant: its activations are the same on every rank (same weights, same
    per-subject priors), so its backward is ONE linear map applied to the gradient of its 25x32^3 output volume.
    Averaging that 3.3 MB volume gradient over the ranks BEFORE the decoder backward (``volume_hook``) gives every
    rank the averaged decoder gradients -- identical to all-reducing them, up to fp32 summation order -- and the
    254 MB never cross xGMI.  What is left (both MLPs and the pose decoder, 3.3 MB) goes through one flat
    persistent bucket after backward (``reduce``).  Per step: two collectives of 3.3 MB, ~0.1 ms on a
    ring over 153 GB/s links, against 2.9 ms for the plain 257.7 MB ring.

    ``mode='full'`` keeps the plain path (decoder gradients in a second flat bucket) for datasets whose priors
    differ between ranks; ``volume`` mode verifies the premise: a checksum of the priors is all-gathered
    every step and compared on the device, the verdict is read back one step late (no host synchronisation) and a
    mismatch raises.

    Parameters that received no gradient (``grad is None``: e.g. the pose decoder before
    ``pose_decoder.kick_in_iter``) stay None, like on one GPU and in the reference: which parameters those are is decided
    by the autograd graph, i.e. by the configuration and ``iter``, identically on every rank; a has-gradient flag per
    parameter rides in the same bucket and is checked (one step late) to be 0 or ``world`` everywhere.
    """

    def __init__(self, network, world, group=None, mode='volume', resync_every=1000, single_rank_collectives=False):
        assert mode in ('volume', 'full')
        self.world, self.group, self.mode = int(world), group, mode
        # world == 1 needs no collective; with ``single_rank_collectives`` they run anyway (over a 1-rank group they are
        # identities): the way to exercise the RCCL code path itself on a box with one GPU (tests/test_gpu_dist.py)
        self.active = self.world > 1 or bool(single_rank_collectives)
        # volume mode: every rank computes the decoder gradients itself from the same averaged volume gradient.  They
        # agree bit for bit as long as the library GEMMs of the decoder backward are run-to-run deterministic; as
        # insurance against replicas drifting apart over 400 k iterations the decoder parameters are re-broadcast from
        # rank 0 every ``resync_every`` steps (254 MB / 1000 steps)
        self.resync_every, self.steps = int(resync_every), 0
        named = [(n, p) for n, p in network.named_parameters() if p.requires_grad]
        self.decoder = [p for n, p in named if 'mweight_vol_decoder' in n]
        self.small = [p for n, p in named if 'mweight_vol_decoder' not in n]
        self.buckets = [self.small] if mode == 'volume' else [self.small, self.decoder]
        self.flat = [None] * len(self.buckets)
        self._flag_cache = [None] * len(self.buckets)      # (has-gradient pattern, device tensor): rebuilt when it changes
        self._pending = []          # [(description, pinned host tensor, event)]
        self.bytes_last_step = 0
        # Replicas must START equal: the volume mode computes the decoder gradients locally on every rank from the same
        # averaged volume gradient (bit-identical decoder weights are its premise), and nothing later would bring
        # differently initialised MLPs together -- a run that does not resume builds each rank's Network from its own
        # random state (train.train_subject seeds with seed + rank).  Rank 0's parameters and buffers win.
        if self.active:
            self.broadcast_state(network)

    def broadcast_state(self, network, src=0):
        """Every parameter and buffer of ``network`` from rank ``src`` (258 MB over xGMI, once)."""
        with torch.no_grad():
            for t in list(network.parameters()) + list(network.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    # -- backward-time hook -------------------------------------------------------------------------------
    def volume_hook(self, vol, priors=None):
        """Wrap the decoded weight volume: its gradient is averaged over the ranks on its way into the decoder."""
        if not self.active or self.mode != 'volume' or not vol.requires_grad:
            return vol
        if priors is not None:
            self._check_same(priors.double().sum().reshape(1), 'motion_weights_priors differ between ranks: the '
                             'decoder activations are not replicated, use cfg.amd.ddp_reduce = "full"')
        self.bytes_last_step += vol.numel() * 4
        return _MeanAllReduceGrad.apply(vol, self)

    # -- after backward -----------------------------------------------------------------------------------
    def reduce(self):
        """Mean-all-reduce the gradients outside the volume hook; in place, None gradients stay None."""
        self._poll()
        if not self.active:
            return
        self.steps += 1
        if self.mode == 'volume' and self.resync_every > 0 and self.steps % self.resync_every == 0:
            for p in self.decoder:
                dist.broadcast(p.data, src=0, group=self.group)
        for bi, bucket in enumerate(self.buckets):
            have = [p for p in bucket if p.grad is not None]
            n = sum(p.numel() for p in bucket)
            if self.flat[bi] is None or self.flat[bi].numel() != n + len(bucket):
                self.flat[bi] = torch.zeros(n + len(bucket), device=bucket[0].device, dtype=torch.float32)
            flat = self.flat[bi]
            views, off = [], 0
            for p in bucket:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            flags = flat[n:]
            if len(have) != len(bucket):
                flat.zero_()
            # has-gradient pattern: it changes a handful of times in a run (kick-in iterations), so its device copy is
            # kept; a fresh pageable host tensor per step would be a stream-synchronising upload per bucket
            pattern = tuple(p.grad is not None for p in bucket)
            if self._flag_cache[bi] is None or self._flag_cache[bi][0] != pattern:
                host = torch.tensor([float(b) for b in pattern])
                if flat.is_cuda:
                    host = host.pin_memory()
                self._flag_cache[bi] = (pattern, host.to(flat.device, non_blocking=True))
            flags.copy_(self._flag_cache[bi][1])
            src = [p.grad for p in have]
            dst = [v for v, p in zip(views, bucket) if p.grad is not None]
            torch._foreach_copy_(dst, src)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat[:n].div_(self.world)
            torch._foreach_copy_(src, dst)
            self.bytes_last_step += flat.numel() * 4
            bad = ((flags != 0) & (flags != self.world)).any().reshape(1)
            self._defer(bad, 'a parameter received a gradient on some ranks only')

    # -- lazy verification (device flag -> pinned host memory, examined on a later step) -----------------------
    def _check_same(self, value, what):
        gathered = [torch.empty_like(value) for _ in range(self.world)]
        dist.all_gather(gathered, value, group=self.group)
        g = torch.stack(gathered)
        self._defer((g != g[0]).any().reshape(1), what)

    def _defer(self, flag, what):
        if flag.is_cuda:
            host = torch.empty(1, dtype=torch.bool, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((what, host, ev))
        elif bool(flag):
            raise RuntimeError(what)

    def _poll(self, wait=False):
        keep = []
        for what, host, ev in self._pending:
            if wait:
                ev.synchronize()
            if ev.query():
                if bool(host):
                    raise RuntimeError(what)
            else:
                keep.append((what, host, ev))
        self._pending = keep

    def finish(self):
        """Examine every outstanding verification flag (end of training / of a test)."""
        self._poll(wait=True)

    def take_bytes(self):
        b, self.bytes_last_step = self.bytes_last_step, 0
        return b


def allreduce_gradients(named_params, world=None):
    """Plain mean-all-reduce of ``.grad`` in place (one collective per tensor list; helper for callers that do not
    use ``GradientSync``).  None gradients are left None."""
    world = world or (dist.get_world_size() if dist.is_initialized() else 1)
    if world == 1:
        return
    grads = [p.grad for _, p in named_params if p.requires_grad and p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
