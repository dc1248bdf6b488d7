"""CPU-side checks of the C ABI: the in-tree library loads and exports every symbol include/pufferlib_amd.h
declares (no compute calls — there is no GPU here), and size helpers behave."""
import ctypes as C
import os
import re

import pytest

from pufferlib_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(REPO, 'include', 'pufferlib_amd.h')).read()
    declared = set(re.findall(r'\b(pfa_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib._SIGNATURES), declared ^ set(_lib._SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name


def test_size_helpers(L):
    assert L.pfa_version() >= 1
    cfg = _lib.SquaredConfig(4096, 3, 1, 64, 64)
    assert L.pfa_squared_state_bytes(C.byref(cfg)) > 4096 * 624 * 4
    bad = _lib.SquaredConfig(0, 3, 1, 64, 64)
    assert L.pfa_squared_state_bytes(C.byref(bad)) == 0
    assert b'num_envs' in L.pfa_last_error()
    dims = _lib.MlpDims(49, 64, 128, 8)
    assert L.pfa_mlp_param_count(C.byref(dims)) == 128 * 64 + 128 + 8 * 128 + 8 + 128 + 1
    assert L.pfa_gae_workspace_bytes(524288) == 256 * 16


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU."""
    import torch
    from pufferlib_amd import vector
    from pufferlib_amd.exceptions import ExtensionError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(ExtensionError):
        vector.make(vector.make_squared, num_envs=4)
