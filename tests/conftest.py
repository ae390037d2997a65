import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def gpu_ctx():
    """One libpgx context for the whole GPU session (fails loudly if the HIP library or
    the device is missing: there is no CPU fallback to fall through to)."""
    # torch brings its own copy of the HIP runtime; when both live in one process, the tests that
    # use torch tensors next to libpgx are reliable only with torch's runtime initialised first
    import torch
    torch.cuda.init()
    from pangenomix_amd import _native
    ctx = _native.Context(0)
    yield ctx
    ctx.close()
