import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(params=['fp32', 'bf16x6'])
def matrix_products(request):
    """Runs a test once per product form of csrc/igemm.hip's rows kernels: the default (exact fp32 MFMA products) and the opt-in
    six-term bf16 split (pfa_igemm_set_products); the default is restored afterwards."""
    from pufferlib_amd import _lib
    L = _lib.lib()
    _lib.check(L.pfa_igemm_set_products(1 if request.param == 'bf16x6' else 0), 'set_products')
    yield request.param
    _lib.check(L.pfa_igemm_set_products(0), 'set_products')
