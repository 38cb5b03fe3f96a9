import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def seeded_params():
    """The seeded 64.4 M-parameter state the golden fixtures were made with."""
    from oracle.seeded import default_shapes, seeded_state
    return seeded_state(default_shapes(), seed=0)


@pytest.fixture(scope='session')
def golden_frame():
    import json
    from humannerf_amd import scene
    with open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')) as f:
        m = json.load(f)['frame']
    return scene.synthetic_frame(H=m['H'], W=m['W'], focal_at_512=m['focal_at_512'],
                                 ray_stride=m['ray_stride'], pose_seed=m['pose_seed'])
