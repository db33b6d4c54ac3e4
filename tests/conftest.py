import os
import sys

import pytest

# The CPU suite runs on a small shared container (8 cores) and spawns 2-rank gloo jobs: cap the BLAS / OpenMP pools so the
# ranks and the parent do not oversubscribe the cores (observed: the same suite taking 90 s or 11 min).  Inherited by
# every child process; must be set before numpy / torch are imported.
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, '4')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def repo_root():
    return ROOT
