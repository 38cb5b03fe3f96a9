"""Golden fixture for the training-time patch sampler (second half): runs the
REFERENCE's Dataset.get_patch_ray_indices (core/data/human_nerf/train.py:236-335, imported read-only, CPU) on a
synthetic ray / subject mask with a seeded global numpy generator and stores what it returns.

    python oracle/make_golden_patches.py        # writes tests/golden/patches_s96.npz

Build container only.  The reference asserts ``mask.dtype == np.bool`` (removed from numpy 1.24+): the alias is
restored for the duration of the call, nothing of the reference is modified or copied."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import GOLD, import_reference            # noqa: E402


def masks(H=96, W=96):
    from humannerf_amd import scene
    fr = scene.synthetic_frame(H=H, W=W, focal_at_512=1250.0)
    yy, xx = np.mgrid[0:H, 0:W]
    # rays that cross the bbox: the frame's own mask cut down to an ellipse, so that windows near its rim are
    # only partly covered (the patch masks then have holes); subject: a narrower ellipse inside it
    rim = (yy - H * 0.5) ** 2 / (H * 0.46) ** 2 + (xx - W * 0.5) ** 2 / (W * 0.30) ** 2 < 1.0
    ray_mask = fr['ray_mask'].astype(bool) & rim.reshape(-1)
    subject = ((yy - H * 0.5) ** 2 / (H * 0.36) ** 2 + (xx - W * 0.5) ** 2 / (W * 0.16) ** 2 < 1.0) & ray_mask.reshape(H, W)
    return ray_mask, subject, ray_mask.reshape(H, W).copy()


def main():
    import types
    cfg, _ = import_reference()
    np.bool = bool                                          # numpy < 1.24 spelling the reference still uses
    # the dataset module imports, at its top, packages that are absent here and that the sampler never calls
    # (same recipe as SURVEY.md section 8(c) step 4): empty stand-ins so that the import succeeds
    for name in ['termcolor', 'imageio', 'tools', 'tools.prepare_zju_mocap', 'tools.prepare_zju_mocap.prepare_dataset']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['termcolor'].colored = lambda s, *a, **k: s
    sys.modules['tools.prepare_zju_mocap.prepare_dataset'].get_mask = None
    from core.data.human_nerf.train import Dataset
    H = W = 96
    ray_mask, subject, bbox = masks(H, W)
    ds = object.__new__(Dataset)                            # the method only reads cfg and its arguments
    np.random.seed(20240)
    sel, info, div = Dataset.get_patch_ray_indices(ds, N_patch=6, ray_mask=ray_mask, subject_mask=subject,
                                                   bbox_mask=bbox, patch_size=20, H=H, W=W)
    np.savez_compressed(os.path.join(GOLD, 'patches_s96.npz'), select_inds=sel, mask=info['mask'],
                        xy_min=info['xy_min'], xy_max=info['xy_max'], div=div,
                        subject_ratio=np.float64(cfg.patch.sample_subject_ratio))
    print('patches:', sel.shape, div, 'subject_ratio', cfg.patch.sample_subject_ratio)


if __name__ == '__main__':
    main()
