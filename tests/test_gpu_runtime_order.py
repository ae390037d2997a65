"""One HIP runtime per process whatever the import order (pangenomix_amd/_native._one_hip_runtime):
a child process loads libpgx BEFORE torch, runs the smoke check, then imports torch, uses the GPU
from torch, and uses libpgx again -- the order that failed when each library brought its own
libamdhip64."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
assert 'torch' not in sys.modules
from pangenomix_amd import _native
ctx = _native.Context(0)                      # libpgx and its HIP runtime first
assert 'torch' not in sys.modules, 'loading libpgx must not import torch'
import __graft_entry__
__graft_entry__.smoke()
import torch
x = torch.arange(1 << 20, device='cuda', dtype=torch.int64)
assert int(x.sum().item()) == (1 << 20) * ((1 << 20) - 1) // 2
import numpy as np
bits = ctx.presence_bitmap(np.array([1, 70], np.int32), np.array([0, 2], np.int32), 100, 3)
assert int(bits[0, 0]) == 2 and int(bits[2, 1]) == 64
runtimes = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln})
assert len(runtimes) == 1, runtimes
print('OK', runtimes[0])
""" % ROOT


def test_libpgx_before_torch_in_one_process():
    out = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.strip().splitlines()[-1].startswith('OK')


def test_torch_before_libpgx_in_one_process():
    child = CHILD.replace("assert 'torch' not in sys.modules\nfrom pangenomix_amd", "import torch\ntorch.cuda.init()\nfrom pangenomix_amd")
    child = child.replace("assert 'torch' not in sys.modules, 'loading libpgx must not import torch'\n", '')
    out = subprocess.run([sys.executable, '-c', child], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
