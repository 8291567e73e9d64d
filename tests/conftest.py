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


@pytest.fixture(scope='session', autouse=True)
def _abort_probe():
    """UPAMD_ABORT_PROBE=<file>: install tests/abort_probe.c's SIGABRT handler for the session (who raised the signal + the C
    backtrace of the raising thread go to <file>).  Used by the round-6 flake loop; off by default."""
    path = os.environ.get('UPAMD_ABORT_PROBE')
    if path:
        import ctypes
        import subprocess
        so = '/tmp/upamd_abort_probe_%d.so' % os.getuid()
        src = os.path.join(ROOT, 'tests', 'abort_probe.c')
        if subprocess.run(['gcc', '-O1', '-g', '-shared', '-fPIC', src, '-o', so], capture_output=True).returncode == 0:
            ctypes.CDLL(so).upamd_abort_probe_install(path.encode())
    yield
