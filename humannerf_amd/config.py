"""Configuration for the MI355X-native HumanNeRF hot path.

The reference keeps a process-global yacs singleton ``cfg`` that every module
imports (reference: configs/config.py:58-80, configs/default.yaml).  The hot
path reads only a handful of its keys *at call time* (SURVEY.md section 5):
``N_samples, perturb, chunk, netchunk_per_gpu, ignore_non_rigid_motions,
total_bones, *.kick_in_iter / full_band_iter`` plus the MLP shapes.

Drop-in rule: when this package is loaded inside the reference tree (its
``configs`` package was already imported by train.py / run.py) we use THAT
object, so late mutations such as ``cfg.perturb = 0.`` made by the reference's
drivers (run.py:71,215; trainer.py:181,265) are honoured.  Stand-alone we use
the defaults below, which restate the default.yaml values the path depends on.
"""
import copy
import sys

import yaml


class CfgNode(dict):
    """Minimal attribute-dict config node (own implementation, not yacs)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value

    def clone(self):
        return copy.deepcopy(self)

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v
        return self

    def merge_from_file(self, path):
        with open(path, 'r') as f:
            return self.merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        """``['a.b', '3', 'c', 'x']`` pairs, values parsed as YAML scalars
        (same convention as the reference CLI, configs/config.py:63)."""
        assert len(opts) % 2 == 0, opts
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f'unknown config key {key}')
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val
        return self


# Values restate configs/default.yaml of the reference (line numbers cited).
_DEFAULTS = {
    'category': 'human_nerf',
    'random_seed': 42,
    'use_amp': False,
    'eval_iter': 10000000,                     # config.py:16
    'ignore_non_rigid_motions': False,         # config.py:20
    'network_module': 'humannerf_amd.network',
    'canonical_mlp': {                         # default.yaml:51-57
        'mlp_depth': 8, 'mlp_width': 256, 'multires': 10, 'i_embed': 0,
        'view_dir': False, 'pose_color': 'wo', 'last_linear_scale': 1,
    },
    'mweight_volume': {                        # default.yaml:133-137
        'embedding_size': 256, 'volume_size': 32, 'dst_voxel_size': 0.0625,
    },
    'posevec': {'type': 'axis_angle'},
    'non_rigid_motion_model': 'mlp',
    'non_rigid_motion_mlp': {                  # default.yaml:142-165
        'condition_code_size': 69, 'pose_input': True, 'time_input': False,
        'mlp_width': 128, 'mlp_depth': 6, 'skips': [4], 'multires': 6,
        'i_embed': 0, 'kick_in_iter': 10000, 'full_band_iter': 50000,
        'last_linear_scale': 1,
    },
    'pose_decoder': {                          # default.yaml:237-242
        'embedding_size': 69, 'mlp_width': 256, 'mlp_depth': 4,
    },
    'pose_decoder_off': False,
    'train': {                                 # default.yaml:261-281
        'perturb': 1.0, 'batch_size': 1, 'shuffle': True, 'drop_last': False,
        'maxiter': 400000, 'lr': 0.0005, 'lr_mweight_vol_decoder': 0.00005,
        'lr_pose_decoder': 0.00005, 'lr_non_rigid_mlp': 0.00005,
        'lrate_decay': 500, 'optimizer': 'adam', 'log_interval': 20,
        'save_checkpt_interval': 2000, 'save_model_interval': 50000,
        'ray_shoot_mode': 'patch', 'selected_frame': 'all',
        'lossweights': {'lpips': 1.0, 'mse': 0.2, 'l1': 0.0},
    },
    'sex': 'neutral',
    'total_bones': 24,                         # default.yaml:346
    'bbox_offset': 0.3,                        # default.yaml:347
    'bgcolor': [0., 0., 0.],
    'patch': {'sample_subject_ratio': 0.8, 'N_patches': 6, 'size': 32},
    'N_samples': 128,                          # default.yaml:357
    'perturb': 1.0,                            # default.yaml:359
    'netchunk_per_gpu': 300000,                # default.yaml:361
    'chunk': 32768,                            # default.yaml:362
    'n_gpus': 1,
    # --- keys that exist only in this build -------------------------------
    'amd': {
        # arithmetic of the two per-sample MLPs (inference): 'f16x3' = fp32-equivalent
        # split-f16 (hi+lo) operands, 3 f16 MFMAs, fp32 accumulate -- passes the same
        # parity tests as 'f32' = v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain),
        # at 3x the speed.
        'mlp_mode': 'f16x3',
        # arithmetic of the activation-saving training forward ('f32' | 'f16x3')
        'train_mlp_mode': 'f16x3',
        # arithmetic of the weight-gradient kernel for the matrix-shaped layers ('f32' | 'f16x3')
        'train_dw_mode': 'f16x3',
        # arithmetic of the two dX chains ('f32' | 'f16x3')
        'train_chain_mode': 'f16x3',
        # storage of the weight-gradient operands between the kernels of a training step (with split-f16 arithmetic in
        # all three places above): 'f16' = activations and dZ travel as f16 (half the HBM bytes of the step's largest
        # buffers; every product of the weight-gradient sums uses 11-bit operands, fp32 accumulation over >= 10^5
        # samples), 'f32' = fp32 storage, 22-bit split operands in the weight-gradient kernel
        'train_operands': 'f16',
        # every this many backward passes the saved activations and the incoming gradient are checked against the
        # range the split-f16 / f16 arithmetic assumes (autograd.OperandRangeGuard); 0 = never
        'train_check_every': 200,
        # inference in 'f16x3' clamps hidden activations at 65504 (the f16 range).  The kernels flag it (status word of
        # the packed weight image, read back one frame late without a synchronisation); what Network.forward does when
        # the flag is up: 'raise' ActivationRangeError | 'f32' = warn and render every later frame with the exact fp32
        # MFMA kernels (the flagged frames are wrong: render loops re-render them) | 'ignore'
        'on_f16_range': 'raise',
        # which launches carry the guard (it costs 3 % of a frame): 'audit' = every ray chunk of the first frame after a
        # weight change, then one rotating chunk per frame | 'full' = every chunk of every frame | 'off'
        'f16_range_guard': 'audit',
        # training items of dataset.FrameStream: patch positions drawn with the reference's exact calls on the global
        # numpy generator (True: 4 ms of host time per item) or from a numpy Generator per item (False: 0.1 ms)
        'exact_patch_draws': False,
        # materialise the per-sample diagnostic outputs the reference always
        # returns (backward_motion_weights, xyz_on_rays, ...; ~17 KB/ray).
        'diagnostics': True,
        # cache the motion-weight volume across eval-mode frames while the
        # decoder parameters and the priors tensor are unchanged.
        'cache_weight_volume': True,
        # lean rendering only: skip the MLPs for samples whose foreground likelihood (sum of
        # skinning weights) is below this; bounds |d rgb|, |d alpha| by ~2 * N_samples * cull_eps.
        # 0 = evaluate every sample exactly like the reference.  1e-9 already drops ~55 % of the
        # samples of a typical frame at an error 100x below the reference's own fp32 noise.
        'cull_eps': 0.0,
        # lean rendering only: stop evaluating a ray once its transmittance is below this (front-to-back slabs of 32
        # samples); bounds |d rgb|, |d alpha| by term_eps.  0 = off (the reference evaluates every sample).
        'term_eps': 0.0,
        # rendering: run the LBS warp (K1) of ray chunk i+1 on a side stream while the MLP kernels of chunk i occupy
        # the main one (ordering by events inside hnrf_render_frame_fwd, no host synchronisation).  Measured on the
        # 512x512x128 frame: 91.5 ms with and without it -- the MLP kernels are power-limited, so whatever K1 draws
        # beside them they lose; off by default
        'overlap_warp': False,
        # multi-GPU training: 'volume' = average the 3.3 MB weight-volume gradient in front of the decoder backward (the
        # decoder's 254 MB of gradients never travel; needs the same priors on every rank, verified at run time),
        # 'full' = plain all-reduce of every gradient
        'ddp_reduce': 'volume',
        # run the collectives of the training step even when world_size == 1 (identities over a one-rank group):
        # exercises the RCCL code path on a single-GPU box (tests/test_gpu_dist.py)
        'ddp_single_rank_collectives': False,
        # gc.freeze() at the start of the training / render loops (config.quiet_gc)
        'freeze_gc': True,
    },
}


_GC_FROZEN = False


def quiet_gc():
    """Take the objects that exist now (modules, the network, the autograd / ctypes machinery: ~10^6 of them) out of the
    cyclic garbage collector's generations.  A full collection otherwise walks all of them every few thousand
    allocations: measured 90-100 ms once per ~138 training steps (0.65 ms per step, 5 %) and the same stall at random
    places in a render loop.  Called once per process by the loops of train.py / render.py and by bench.py; the collector
    stays enabled for what is allocated afterwards.  cfg.amd.freeze_gc = False turns it off."""
    global _GC_FROZEN
    if _GC_FROZEN or not amd_option('freeze_gc', True):
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_FROZEN = True


def get_cfg_defaults():
    return CfgNode(_DEFAULTS)


def _resolve_cfg():
    ref = sys.modules.get('configs')
    if ref is not None and hasattr(ref, 'cfg'):
        return ref.cfg           # reference singleton: drop-in mode
    return get_cfg_defaults()


cfg = _resolve_cfg()


def amd_option(name, default=None):
    """Read a build-specific option; the reference cfg has no ``amd`` node."""
    node = cfg.get('amd', None) if hasattr(cfg, 'get') else None
    if node is None:
        return _DEFAULTS['amd'].get(name, default)
    return node.get(name, _DEFAULTS['amd'].get(name, default))
