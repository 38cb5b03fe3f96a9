"""Drop-in at the reference's plugin boundary (core/nets/create_network.py:6-15): the reference
loads ``cfg.network_module`` with imp.load_source and calls ``Network()``.  Runs only where the
reference tree is mounted (the build container); skipped on the GPU box.  Executed in a subprocess
because the reference's ``configs`` package parses sys.argv at import and installs a global cfg."""
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import imp, os, sys, types
sys.dont_write_bytecode = True
REF, ROOT = %r, %r
os.makedirs('/tmp/hnrf_oracle', exist_ok=True)
open('/tmp/hnrf_oracle/dropin.yaml', 'w').write(
    "task: 'zju_mocap'\nsubject: 'p387'\nexperiment: 'dropin'\nprimary_gpus: ['cpu']\nsecondary_gpus: ['cpu']\n"
    "bgcolor: [0., 0., 0.]\nresize_img_scale: 0.5\nnetwork_module: 'humannerf_amd.network'\n")
os.chdir(REF); sys.path.insert(0, REF); sys.path.insert(1, ROOT)
sys.argv = ['x', '--cfg', '/tmp/hnrf_oracle/dropin.yaml']
for name in ['cv2', 'torchvision', 'torchvision.models', 'torchvision.transforms']:
    sys.modules[name] = types.ModuleType(name)
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.modules['torchvision.transforms'].Compose = lambda *a, **k: None
sys.modules['torchvision.transforms'].Normalize = lambda *a, **k: None
import torch
from configs import cfg                       # the reference's singleton
assert cfg.network_module == 'humannerf_amd.network'
# what create_network() does, with the path resolved against this repo instead of the cwd
mine = imp.load_source(cfg.network_module, os.path.join(ROOT, cfg.network_module.replace('.', '/') + '.py')).Network()
from humannerf_amd import config as hcfg
assert hcfg.cfg is cfg, 'the build must use the reference cfg singleton when it is loaded'
ref = imp.load_source('core.nets.human_nerf.network', 'core/nets/human_nerf/network.py').Network()
a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
assert a == b, set(a) ^ set(b)
mine.load_state_dict(ref.state_dict(), strict=True)          # a reference checkpoint loads unchanged
assert mine.deploy_mlps_to_secondary_gpus() is mine
# optimizer routing of the reference works on our parameter names
from core.train.optimizers.human_nerf.optimizer import get_optimizer
opt = get_optimizer(mine)
names = {g['name'] for g in opt.param_groups}
assert {'mweight_vol_decoder', 'pose_decoder', 'non_rigid_mlp'} <= names
# late mutation of the shared cfg is what the hot path reads
cfg.perturb = 0.
assert hcfg.cfg.perturb == 0.
print('DROPIN_OK', len(a))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted (GPU box)')
def test_network_is_a_drop_in_for_the_reference_factory():
    r = subprocess.run([sys.executable, '-c', SCRIPT % (REF, ROOT)], capture_output=True, text=True, timeout=600)
    assert 'DROPIN_OK 55' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
