import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: full-size GPU checks that take tens of seconds each')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def gpu_ctx():
    """One libpgx context for the whole GPU session (fails loudly if the HIP library or
    the device is missing: there is no CPU fallback to fall through to)."""
    from pangenomix_amd import _native     # (maps ONE HIP runtime for the process, see _native._one_hip_runtime)
    ctx = _native.Context(0)
    yield ctx
    ctx.close()
