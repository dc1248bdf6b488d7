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


def test_weight_gradient_workspace_never_shrinks_with_more_rows(L):
    """cnn.Engine / general.Engine size the dW workspace once for their largest chunk and reuse it for shorter ones: the launch plan
    of csrc/igemm.hip (row splits x partial tiles) must therefore never need MORE room for FEWER rows — also where a small gradient
    is split finer (>= 256 rows per split until ~1024 workgroups exist)."""
    shapes = [(64, 256), (256, 16), (512, 16), (256, 32), (512, 64), (576, 64), (3136, 512), (160, 128), (48, 64), (1024, 512)]
    for K, N in shapes:
        prev = 0
        for M in (1, 16, 255, 256, 257, 4096, 8192, 65536, 131072, 8192 * 49, 8192 * 81, 8192 * 400):
            b = L.pfa_igemm_weights_workspace_bytes(M, K, N)
            assert b >= prev > -1, (K, N, M, b, prev)
            assert b >= K * N * 4, (K, N, M)
            prev = b
    assert L.pfa_igemm_weights_workspace_bytes(0, 64, 64) == 0 and L.pfa_igemm_weights_workspace_bytes(100, 64, 24) == 0
    # gemm_tn (csrc/gemm.hip): the 160-float encoder gradient has its own strip
    assert L.pfa_gemm_tn_workspace_bytes(128, 160, 131072) > 0 and L.pfa_gemm_tn_workspace_bytes(128, 24, 4096) == 0


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU."""
    import torch
    from pufferlib_amd import vector
    from pufferlib_amd.exceptions import ExtensionError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(ExtensionError):
        vector.make(vector.make_squared, num_envs=4)
