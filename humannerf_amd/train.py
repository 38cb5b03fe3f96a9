"""Training step around the hot path ("next" row).

Mirrors what the reference's trainer does per iteration (core/train/trainers/
human_nerf/trainer.py:186-255), restated for one-process-per-GPU data parallelism:

  * optimizer: Adam(betas=(0.9, 0.999)), one param group per tensor, learning rate
    routed by name substring from cfg.train.lr_* (optimizers/human_nerf/optimizer.py:12-43);
  * learning-rate schedule: base * 0.1 ** (iter / (lrate_decay * 1000))
    (lr_updaters/exp_decay.py:7-16);
  * loss: patches rebuilt from the rendered rays by mask / div indices with background fill
    (trainer.py:28-37), 0.2 * MSE (+ 1.0 * LPIPS in the reference; the VGG trunk cannot be
    fetched offline, so LPIPS is a pluggable callable that defaults to absent);
  * every rank renders its own frame; gradients are mean-all-reduced over RCCL in two flat
    buckets (humannerf_amd/dist.py) before the optimizer step.  The reference's
    nn.DataParallel uses ONE frame per step whatever the GPU count: with N ranks the
    effective batch here is N frames (stated with every iters/s number);
  * checkpoint dict {'iter', 'network', 'optimizer'} (trainer.py:356-364).
"""
import torch

from . import dist as hdist
from .config import cfg


def customized_lr_names():
    return [k[3:] for k in cfg.train.keys() if k.startswith('lr_')]


def build_optimizer(network):
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
    # same per-parameter groups as the reference (optimizer checkpoints stay interchangeable); on the GPU the
    # update of each group is one fused kernel instead of torch's default chain of foreach kernels
    fused = all(g['params'][0].is_cuda for g in groups)
    return torch.optim.Adam(groups, lr=cfg.train.lr, betas=(0.9, 0.999), fused=fused)


def update_lr(optimizer, iter_step):
    decay = 0.1 ** (iter_step / (cfg.train.lrate_decay * 1000))
    for group in optimizer.param_groups:
        base = cfg.train.get('lr_' + str(group['name']), cfg.train.lr)
        group['lr'] = base * decay


def unpack_patches(rgbs, patch_masks, bgcolor, targets, div_indices):
    """(sum_rays, 3) rendered colours -> (N_patch, H, W, 3) images, background elsewhere."""
    n_patch = len(div_indices) - 1
    assert patch_masks.shape[0] == n_patch and targets.shape[0] == n_patch
    imgs = bgcolor.expand(targets.shape).clone()
    for i in range(n_patch):
        imgs[i, patch_masks[i]] = rgbs[int(div_indices[i]):int(div_indices[i + 1])]
    return imgs


def image_loss(rgb, target, lpips_fn=None):
    """sum_k lossweights[k] * loss_k over the weights > 0 (trainer.py:97-175, single-head branch)."""
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


class Trainer:
    def __init__(self, network, optimizer=None, lpips_fn=None, world_size=1):
        self.network = network.deploy_mlps_to_secondary_gpus()
        self.optimizer = optimizer if optimizer is not None else build_optimizer(network)
        self.lpips_fn = lpips_fn
        self.world_size = world_size
        self.iter = 1

    def train_step(self, batch):
        """One optimizer step on one frame of this rank (trainer.py:200-231)."""
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
        hdist.allreduce_gradients(self.network.named_parameters(), self.world_size)
        self.optimizer.step()
        update_lr(self.optimizer, self.iter)
        self.iter += 1
        return loss.detach(), {k: v.detach() for k, v in parts.items()}

    def state(self):
        return {'iter': self.iter, 'network': self.network.state_dict(), 'optimizer': self.optimizer.state_dict()}

    def load_state(self, ckpt):
        self.iter = ckpt['iter'] + 1
        self.network.load_state_dict(ckpt['network'], strict=False)
        self.optimizer.load_state_dict(ckpt['optimizer'])
