import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def repo_root():
    return ROOT


@pytest.fixture(scope='session', autouse=True)
def _lab_thread_variant():
    """UPAMD_TEST_TINY_THREADS=512 runs the whole GPU suite on the 512-thread variant of the fused small-model kernel (the default
    is 1024; tests that set the knob themselves put it back to the value named here: test_gpu_parity.tune)."""
    v = os.environ.get('UPAMD_TEST_TINY_THREADS')
    if v:
        import torch
        if torch.cuda.is_available():
            from drl_urban_planning_amd import native
            native.check(native.lib().upamd_tune(b'tiny_threads', int(v)), 'upamd_tune')
    yield
