"""Training step and loop around the hot path ("next" row).

Mirrors what the reference's trainer does per iteration (core/train/trainers/
human_nerf/trainer.py:186-255), restated for one-process-per-GPU data parallelism:

  * optimizer: Adam(betas=(0.9, 0.999)), one param group per tensor, learning rate
    routed by name substring from cfg.train.lr_* (optimizers/human_nerf/optimizer.py:12-43).
    ``GroupedAdam`` keeps exactly that group structure (``state_dict()`` is interchangeable with the
    reference's optimizer checkpoints) but updates all groups that share hyper-parameters in ONE
    multi-tensor launch -- two launches per step (5e-4 and 5e-5) instead of 56;
  * learning-rate schedule: base * 0.1 ** (iter / (lrate_decay * 1000))
    (lr_updaters/exp_decay.py:7-16), applied after the step like trainer.py:253;
  * loss: patches rebuilt from the rendered rays by mask / div indices with background fill
    (trainer.py:28-37), 0.2 * MSE (+ 1.0 * LPIPS in the reference; the VGG trunk cannot be
    fetched offline, so LPIPS is a pluggable callable; without it the objective is MSE-ONLY and
    the trainer says so once -- ``Trainer.objective``);
  * every rank renders its own frame; gradients are averaged over RCCL by ``dist.GradientSync``
    (volume-gradient all-reduce in front of the decoder backward + one 3.3 MB bucket).  The reference's
    nn.DataParallel uses ONE frame per step whatever the GPU count: with N ranks the
    effective batch here is N frames (stated with every iters/s number);
  * checkpoints: dict {'iter', 'network', 'optimizer'} in ``<logdir>/<name>.tar`` (trainer.py:356-377),
    read back with ``weights_only=True``; 'latest' every cfg.train.save_checkpt_interval iterations and at the
    first one, 'iter_N' every cfg.train.save_model_interval when cfg.save_all (trainer.py:246-251);
  * progress renders at iterations {start, 100, 300, 1000, 2500} and every cfg.progress.dump_interval
    (trainer.py:240-243, 271-350) through a caller-supplied callable.
"""
import os
import warnings

import torch

from . import dist as hdist
from .config import cfg, amd_option, quiet_gc


def customized_lr_names():
    return [k[3:] for k in cfg.train.keys() if k.startswith('lr_')]


class GroupedAdam(torch.optim.Adam):
    """torch.optim.Adam with the reference's one-group-per-tensor layout, stepped with one fused multi-tensor
    launch per distinct (lr, betas, eps, weight_decay) instead of one per group.  Same update rule as
    torch's fused Adam (bias-corrected, eps outside the square root); parameters whose grad is None are skipped and
    their step counters do not advance."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)

    def _compact_params(self):
        """{index in the saved state: parameter} of the parameters that hold only the central taps of a reference
        tensor (network._PointDeconv: ``hnrf_full_shape``)."""
        out, idx = {}, 0
        for group in self.param_groups:
            for p in group['params']:
                if getattr(p, 'hnrf_full_shape', None) is not None:
                    out[idx] = p
                idx += 1
        return out

    def state_dict(self):
        """The reference's layout (trainer.py:356-364): moments of a compact parameter are written in the full tensor's
        shape, zero outside the central taps -- which is what they are in the reference's own checkpoints."""
        from .network import expand_central_taps
        sd = super().state_dict()
        for idx in self._compact_params():
            st = sd['state'].get(idx)
            if st:
                sd['state'][idx] = dict(st, exp_avg=expand_central_taps(st['exp_avg']),
                                        exp_avg_sq=expand_central_taps(st['exp_avg_sq']))
        return sd

    def load_state_dict(self, state_dict):
        """Also accepts what the REFERENCE's optimizer wrote (trainer.py:356-364: a plain torch.optim.Adam -- groups with
        ``fused`` None / False and per-parameter ``step`` counters that are Python numbers or CPU tensors).  torch
        restores groups and step placement as saved; the fused multi-tensor launch of ``step`` needs every counter as a
        float32 scalar ON the parameter's device, and the groups marked fused."""
        compact = self._compact_params()
        if compact:
            state_dict = dict(state_dict, state=dict(state_dict['state']))
            for idx, p in compact.items():
                st = state_dict['state'].get(idx)
                if st and tuple(st['exp_avg'].shape) == tuple(p.hnrf_full_shape):
                    state_dict['state'][idx] = dict(st, exp_avg=st['exp_avg'][:, :, 1:3, 1:3, 1:3].contiguous(),
                                                    exp_avg_sq=st['exp_avg_sq'][:, :, 1:3, 1:3, 1:3].contiguous())
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group['fused'], group['foreach'] = True, False
            group.setdefault('capturable', False)
            group.setdefault('differentiable', False)
            for p in group['params']:
                st = self.state.get(p)
                if st and 'step' in st:
                    st['step'] = torch.as_tensor(float(st['step']), dtype=torch.float32).to(p.device)
                    for k in ('exp_avg', 'exp_avg_sq'):
                        st[k] = st[k].to(device=p.device, dtype=p.dtype)

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        classes = {}
        for group in self.param_groups:
            assert not group['amsgrad'] and not group['maximize']
            key = (float(group['lr']), tuple(group['betas']), float(group['eps']), float(group['weight_decay']))
            cls = classes.setdefault(key, ([], [], [], [], []))
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                cls[0].append(p)
                cls[1].append(p.grad)
                cls[2].append(st['exp_avg'])
                cls[3].append(st['exp_avg_sq'])
                cls[4].append(st['step'])
        for (lr, (b1, b2), eps, wd), (ps, gs, m, v, steps) in classes.items():
            if not ps:
                continue
            torch._foreach_add_(steps, 1)
            torch._fused_adam_(ps, gs, m, v, [], steps, amsgrad=False, lr=lr, beta1=b1, beta2=b2, weight_decay=wd,
                               eps=eps, maximize=False, grad_scale=None, found_inf=None)
            # torch._fused_adam_ called directly does NOT advance the parameters' version counters (the optimizer wrapper's
            # bookkeeping is what normally does) -- and the network's packed-weight and weight-volume caches are keyed by
            # them: without this an eval render after training steps (Trainer's progress mosaics, a render at the end of a
            # run) silently used the weights of before.  An in-place no-op on an empty view bumps the shared counter
            # without a launch.
            for p in ps:
                p.detach().reshape(-1)[:0].zero_()


def build_optimizer(network):
    """optimizers/human_nerf/optimizer.py:12-43."""
    groups = []
    names = customized_lr_names()
    for key, value in network.named_parameters():
        if not value.requires_grad:
            continue
        hit = [n for n in names if n in key]
        if hit:
            groups += [{'params': [value], 'lr': cfg.train['lr_' + n], 'name': n} for n in hit]
        else:
            groups.append({'params': [value], 'name': key})
    if cfg.train.optimizer != 'adam':
        raise ValueError('Unsupported optimizer ' + str(cfg.train.optimizer))
    return GroupedAdam(groups, lr=cfg.train.lr, betas=(0.9, 0.999))


def update_lr(optimizer, iter_step):
    """lr_updaters/exp_decay.py:7-16."""
    decay = 0.1 ** (iter_step / (cfg.train.lrate_decay * 1000))
    for group in optimizer.param_groups:
        base = cfg.train.get('lr_' + str(group['name']), cfg.train.lr)
        group['lr'] = base * decay


def unpack_patches(rgbs, patch_masks, bgcolor, targets, div_indices):
    """(sum_rays, 3) rendered colours -> (N_patch, H, W, 3) images, background elsewhere (trainer.py:28-37).  The rays
    arrive patch after patch, row-major inside a patch -- the order in which a boolean index walks ``patch_masks`` -- so
    one masked scatter places them all; a boolean-index assignment per patch (the reference's loop) would cost a
    device synchronisation each."""
    n_patch = len(div_indices) - 1
    assert patch_masks.shape[0] == n_patch and targets.shape[0] == n_patch
    assert int(div_indices[-1]) == rgbs.shape[0]
    imgs = bgcolor.expand(targets.shape).clone()
    return imgs.masked_scatter_(patch_masks[..., None].expand_as(imgs), rgbs)


def image_loss(rgb, target, lpips_fn=None):
    """sum_k lossweights[k] * loss_k over the weights > 0 (trainer.py:97-175, single-head branch).  The LPIPS term
    is left out when no ``lpips_fn`` is given (see Trainer.objective)."""
    weights = {k: v for k, v in cfg.train.lossweights.items() if v > 0}
    total, parts = 0.0, {}
    if 'mse' in weights:
        parts['mse'] = weights['mse'] * torch.mean((rgb - target) ** 2)
    if 'l1' in weights:
        parts['l1'] = weights['l1'] * torch.mean(torch.abs(rgb - target))
    if 'lpips' in weights and lpips_fn is not None:
        parts['lpips'] = weights['lpips'] * torch.mean(lpips_fn(rgb.permute(0, 3, 1, 2) * 2. - 1.,
                                                               target.permute(0, 3, 1, 2) * 2. - 1.))
    for v in parts.values():
        total = total + v
    return total, parts


PROGRESS_ITERS = (100, 300, 1000, 2500)          # trainer.py:240


class Trainer:
    def __init__(self, network, optimizer=None, lpips_fn=None, world_size=1, process_group=None, logdir=None):
        self.network = network.deploy_mlps_to_secondary_gpus()
        self.optimizer = optimizer if optimizer is not None else build_optimizer(network)
        self.lpips_fn = lpips_fn
        self.world_size = int(world_size)
        self.logdir = logdir if logdir is not None else cfg.get('logdir', None)
        self.grad_sync = hdist.GradientSync(network, self.world_size, group=process_group,
                                            mode=amd_option('ddp_reduce', 'volume'),
                                            single_rank_collectives=amd_option('ddp_single_rank_collectives', False))
        network.grad_sync = self.grad_sync if self.grad_sync.active else None
        weights = {k: v for k, v in cfg.train.lossweights.items() if v > 0}
        self.objective = ' + '.join('%g*%s' % (v, k) for k, v in weights.items() if k != 'lpips' or lpips_fn is not None)
        if 'lpips' in weights and lpips_fn is None:
            # the reference objective is 1.0*LPIPS + 0.2*MSE (default.yaml:278-281); the VGG trunk is a remote fetch
            warnings.warn('cfg.train.lossweights.lpips = %g but no lpips_fn was supplied: training on %s only '
                          '(set lossweights.lpips = 0 to silence this)' % (weights['lpips'], self.objective))
        self.iter = 1
        self.start_iter = 1
        quiet_gc()

    # ------------------------------------------------------------------------------------------ one iteration
    def backward_step(self, batch):
        """Forward, loss, backward and the gradient averaging of one iteration (trainer.py:200-218): leaves the
        (rank-averaged) gradients in ``.grad``."""
        self.network.train()
        self.optimizer.zero_grad(set_to_none=True)
        out = self.network(**batch, iter_val=float(self.iter))
        if 'patch_masks' in batch:
            pred = unpack_patches(out['rgb'], batch['patch_masks'], batch['bgcolor'] / 255.,
                                  batch['target_patches'], batch['patch_div_indices'])
            loss, parts = image_loss(pred, batch['target_patches'], self.lpips_fn)
        else:                                # flat rays with per-ray targets
            loss, parts = image_loss(out['rgb'][None, None], batch['target_rgbs'][None, None], None)
        loss.backward()
        self.grad_sync.reduce()
        return loss.detach(), {k: v.detach() for k, v in parts.items()}

    def optimizer_step(self):
        """Adam update, learning-rate decay, iteration counter (trainer.py:219, 253-255)."""
        self.optimizer.step()
        update_lr(self.optimizer, self.iter)
        self.iter += 1

    def train_step(self, batch):
        """One optimizer step on one frame of this rank."""
        res = self.backward_step(batch)
        self.optimizer_step()
        return res

    # ------------------------------------------------------------------------------------------ the loop
    def train(self, batches, maxiter=None, progress_fn=None, log_fn=print, rank=0):
        """trainer.py:186-255 over an iterable of per-frame batches (already on the device).  ``progress_fn(trainer)``
        renders the progress frames; checkpoints are written by rank 0 only."""
        maxiter = int(cfg.train.maxiter if maxiter is None else maxiter)
        quiet_gc()
        old = cfg.perturb
        cfg.perturb = cfg.train.perturb                                   # trainer.py:181
        try:
            for batch in batches:
                if self.iter > maxiter:
                    break
                it = self.iter
                loss, parts = self.train_step(batch)
                if it % cfg.train.log_interval == 0 and log_fn is not None:
                    log_fn('Iter %d  Loss: %.4f [%s]' % (it, float(loss), ' '.join('%s: %.4f' % (k, float(v))
                                                                                   for k, v in parts.items())))
                dump = cfg.get('progress', {}).get('dump_interval', 5000)
                if progress_fn is not None and (it == self.start_iter or it in PROGRESS_ITERS or it % dump == 0):
                    self.iter = it                                        # progress renders with the iteration just run
                    was = cfg.perturb
                    cfg.perturb = 0.                                      # trainer.py:263-269
                    self.network.eval()
                    try:
                        progress_fn(self)
                    finally:
                        self.network.train()
                        cfg.perturb = was
                        self.iter = it + 1
                if rank == 0 and self.logdir is not None:
                    self.iter = it                                        # checkpoints carry the iteration just run
                    try:
                        if it % cfg.train.save_checkpt_interval == 0 or it == self.start_iter:
                            self.save_ckpt('latest')
                        if cfg.get('save_all', False) and it % cfg.train.save_model_interval == 0:
                            self.save_ckpt('iter_%d' % it)
                    finally:
                        self.iter = it + 1
        finally:
            cfg.perturb = old
            self.grad_sync.finish()

    # ------------------------------------------------------------------------------------------ checkpoints
    def state(self):
        return {'iter': self.iter, 'network': self.network.state_dict(), 'optimizer': self.optimizer.state_dict()}

    def load_state(self, ckpt):
        """trainer.py:366-377.  Like the reference, the network is loaded non-strictly, but what did not match is
        reported instead of silently dropped."""
        self.iter = int(ckpt['iter']) + 1
        self.start_iter = self.iter
        res = self.network.load_state_dict(ckpt['network'], strict=False)
        if res.missing_keys or res.unexpected_keys:
            warnings.warn('checkpoint does not match the network: missing %s, unexpected %s'
                          % (sorted(res.missing_keys), sorted(res.unexpected_keys)))
        self.optimizer.load_state_dict(ckpt['optimizer'])
        return res

    def ckpt_path(self, name):
        return name if os.path.isfile(name) else os.path.join(self.logdir, '%s.tar' % name)

    def save_ckpt(self, name):
        os.makedirs(self.logdir, exist_ok=True)
        path = os.path.join(self.logdir, '%s.tar' % name)
        torch.save(self.state(), path)
        return path

    def load_ckpt(self, name, map_location=None):
        ckpt = load_checkpoint(self.ckpt_path(name), map_location=map_location)
        return self.load_state(ckpt)


def make_progress_fn(frames, logdir, device=None, imgs_per_row=4):
    """The reference's Trainer.progress (trainer.py:271-350) as a ``progress_fn`` for Trainer.train: renders ``frames``
    (per-frame dicts like the 'progress' dataset yields: 16 frames in image mode, each with ``target_rgbs``) with the
    current weights, puts render and truth side by side, tiles them and writes ``prog_{iter:06}.jpg`` into ``logdir``.
    Returns True when a rendered image is empty (all background) during the first 5000 iterations, like the reference."""
    import numpy as np
    from . import render

    def progress(trainer):
        os.makedirs(logdir, exist_ok=True)
        pairs, empty = {}, False

        def on_image(i, rgb8, a8, t8=None):
            pairs[i] = np.concatenate([rgb8, t8 if t8 is not None else a8], axis=1)

        old_iter = cfg.get('eval_iter', None)
        cfg.eval_iter = trainer.iter                                   # progress renders use the CURRENT iteration (trainer.py:293)
        try:
            imgs = render.render_frames(trainer.network, frames, device=device, on_image=on_image, show_truth=True)
        finally:
            cfg.eval_iter = old_iter
        if trainer.iter <= 5000:
            bg = np.array(cfg.bgcolor, dtype=np.float32)
            empty = any(np.allclose(im, bg, atol=5.) for im in imgs.values())
        from PIL import Image
        tiled = render.tile_images([pairs[i] for i in sorted(pairs)], imgs_per_row=imgs_per_row)
        Image.fromarray(tiled).save(os.path.join(logdir, 'prog_%06d.jpg' % trainer.iter))
        return empty
    return progress


def train_subject(network, subject, logdir, rank=0, world=1, device=None, maxiter=None, lpips_fn=None, seed=None,
                  progress=True, log_fn=print):
    """What the reference's train.py main() does (train.py:17-40, trainer.py:46-75, 177-184) for one prepared subject
    directory, one process per GPU: resume from ``<logdir>/<cfg.load_net>.tar`` when cfg.resume is set and it exists
    (otherwise save 'init'), stream shuffled frames of this rank's shard (dataset.FrameStream), step until
    cfg.train.maxiter with the progress mosaic of 16 evenly spaced frames (create_dataset.py:47-52) and the
    checkpoint cadence of Trainer.train, save 'latest' at the end (trainer.py:352-354).  Returns the Trainer."""
    from . import dataset
    device = device or next(network.parameters()).device
    seed = int(cfg.get('random_seed', 0) if seed is None else seed)
    torch.manual_seed(seed + rank)
    trainer = Trainer(network, lpips_fn=lpips_fn, world_size=world, logdir=logdir)
    name = str(cfg.get('load_net', 'latest'))
    if cfg.get('resume', False) and os.path.isfile(os.path.join(logdir, name + '.tar')):
        trainer.load_ckpt(name, map_location=device)
    elif rank == 0:
        trainer.iter = 0
        trainer.save_ckpt('init')
        trainer.iter = 1
    progress_fn = None
    if progress and rank == 0:
        total = len(subject.framelist_all)
        prog = dataset.Subject(subject.dataset_path, skip=max(1, total // 16), maxframes=16)
        frames = [prog.movement_frame(i, load_image=True) for i in range(len(prog))]
        progress_fn = make_progress_fn(frames, logdir, device=device)
    stream = dataset.FrameStream(subject, rank=rank, world=world, seed=seed, device=device)
    try:
        trainer.train(stream, maxiter=maxiter, progress_fn=progress_fn, log_fn=log_fn if rank == 0 else None, rank=rank)
    finally:
        stream.close()
    if rank == 0:
        trainer.save_ckpt('latest')                                 # Trainer.finalize (trainer.py:257-258)
    return trainer


def load_checkpoint(path, map_location=None):
    """Read a ``.tar`` checkpoint of the reference's layout ({'iter', 'network', 'optimizer'}, trainer.py:356-364)
    without executing anything from the file (``weights_only=True``)."""
    ckpt = torch.load(path, map_location=map_location or 'cpu', weights_only=True)
    if not isinstance(ckpt, dict) or 'network' not in ckpt:
        raise ValueError('%s is not a HumanNeRF checkpoint (keys: %s)' % (path, list(ckpt)[:8] if isinstance(ckpt, dict) else type(ckpt)))
    return ckpt


def load_network(network, path, map_location=None):
    """run.py:18-34: load ``ckpt['network']`` into ``network`` (non-strict like the reference), reporting every
    key that did not match."""
    ckpt = load_checkpoint(path, map_location=map_location)
    res = network.load_state_dict(ckpt['network'], strict=False)
    if res.missing_keys or res.unexpected_keys:
        warnings.warn('checkpoint does not match the network: missing %s, unexpected %s'
                      % (sorted(res.missing_keys), sorted(res.unexpected_keys)))
    return res
