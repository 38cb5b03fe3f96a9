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
    from humannerf_amd.seeded import default_shapes, seeded_state
    return seeded_state(default_shapes(), seed=0)


@pytest.fixture(scope='session')
def golden_meta():
    import json
    with open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_frame(golden_meta):
    """The default golden frame (cases may override parts of it: golden_case)."""
    from humannerf_amd import scene
    m = {k: v for k, v in golden_meta['frame'].items() if k != 'seed'}
    return scene.synthetic_frame(**m)


@pytest.fixture(scope='session')
def golden_case(golden_meta, seeded_params):
    """case name -> (meta, npz, frame, state): the inputs oracle/make_golden.py fed the reference for that case."""
    import numpy as np
    from humannerf_amd import scene
    from humannerf_amd.seeded import with_density
    frames = {}

    def get(case):
        m = golden_meta[case]
        key = repr(sorted(m['frame'].items()))
        if key not in frames:
            frames[key] = scene.synthetic_frame(**m['frame'])
        state = seeded_params if m['density'] is None else with_density(seeded_params, *m['density'])
        return m, np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz')), frames[key], state
    return get
