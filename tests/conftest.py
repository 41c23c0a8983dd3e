import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built library (it is git-ignored): build it once, in-tree, before any test imports it
    lib = os.path.join(ROOT, 'syconn_amd', 'libsyconn_dense_hip.so')
    if not os.path.isfile(lib) and os.path.isfile('/opt/rocm/bin/hipcc'):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('this test is marked gpu but no ROCm device is visible')
    return torch.device('cuda', 0)
