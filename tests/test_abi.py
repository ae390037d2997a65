"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/pgx.h declares; compute entry points refuse to run without a device (no CPU
fallback)."""
import ctypes
import os
import re

import pytest

from pangenomix_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'pgx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pgx_[a-z_0-9]+)\s*\(', text)))


def test_header_declares_what_the_binding_binds():
    assert declared_functions() == sorted(_native.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, 'pangenomix_amd', 'libpgx.so'))
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert _native.lib().pgx_version() == 300


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_native.ClusterParams) == 6 * 4 + 3 * 8 + 2 * 4 + 4 * 8 + 8 + 2 * 4
    assert ctypes.sizeof(_native.ClusterStats) == 16 * 8
    assert ctypes.sizeof(_native.DeviceInfo) == 64 + 32 + 4 * 4 + 8 + 2 * 4


def test_stride_is_a_multiple_of_128_bytes():
    f = _native.lib().pgx_bitmap_stride_words
    assert [f(n) for n in (0, 1, 64, 65, 1024, 1025, 150000)] == [16, 16, 16, 16, 16, 32, 2352]


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(_native.PgxError, match='no usable HIP device|no CPU fallback'):
        _native.Context(0)


def test_rccl_is_loaded_on_request_only_and_a_bad_path_is_an_error():
    """libpgx does not link RCCL: the collective library is dlopen'ed by path (pgx_rccl_load). A path that does not
    exist is PGX_ERR_INVALID with the loader's message, and without a communicator a sharded call is refused."""
    with pytest.raises(_native.PgxError, match='pgx_rccl_load'):
        _native.rccl_load('/nonexistent/librccl.so')
    needed = os.popen('readelf -d %s' % os.path.join(ROOT, 'pangenomix_amd', 'libpgx.so')).read()
    assert 'rccl' not in needed
    assert os.path.basename(_native.rccl_path()).startswith('librccl')
