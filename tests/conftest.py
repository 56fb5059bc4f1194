import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the shipped MIOpen find-db + kernel cache (geoformer_amd/miopen.py) before the first convolution of the session: the training tests'
    # first backward otherwise spends minutes inside MIOpen on a fresh box.  It only sets two environment variables and copies small files.
    try:
        from geoformer_amd import miopen
        miopen.use_shipped_find_db()
    except Exception:                      # (a broken import surfaces in the tests that need the package)
        pass


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load
