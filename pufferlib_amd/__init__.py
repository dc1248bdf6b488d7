"""pufferlib_amd — MI355X (gfx950) native PPO rollout-and-update engine behind PufferLib's own surfaces.

Only the hot path named in BASELINE.json is here:

  pufferlib_amd.vector        make() + the device-resident Squared backend (pufferlib/vector.py protocol)
  pufferlib_amd.clean_pufferl create / evaluate / train / close (clean_pufferl.py surface)
  pufferlib_amd.models        Default (models.py:12-62) as a parameter container over one flat device buffer
  pufferlib_amd.cleanrl       Policy wrapper (frameworks/cleanrl.py:50-66)
  pufferlib_amd.csrc          hand-written HIP kernels + the C ABI declared in include/pufferlib_amd.h

All compute goes through libpufferlib_amd.so (HIP).  There is no CPU fallback: importing the kernels
without the built extension, or calling them without a GPU, raises.
"""
import os as _os
import sys as _sys

from .namespace import namespace, Namespace  # noqa: F401


def _ipc_mode_guard():
    """The peer-mapped exchange of the data-parallel update (csrc/p2p.hip) shares device memory between processes with dmabuf IPC,
    which the HSA runtime only offers when HSA_ENABLE_IPC_MODE_LEGACY=0 was in the environment WHEN HIP INITIALISED.  So the variable
    is set at import of this package (before anything of it touches the device), and what was found is recorded: a process that had
    already initialised HIP without it cannot be fixed any more — dist.init_p2p then leaves the peer path closed (the update runs on
    RCCL / torch.distributed) and says why, instead of discovering it in the self-test."""
    had = _os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')
    torch = _sys.modules.get('torch')
    hip_up = False
    try:
        hip_up = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        pass
    if had is None:
        _os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    ok = had == '0' or (had is None and not hip_up)
    reason = None
    if not ok:
        reason = (f'HSA_ENABLE_IPC_MODE_LEGACY={had} in the environment (the peer path needs 0)' if had is not None else
                  'HIP was initialised before pufferlib_amd was imported, without HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment')
    return dict(value_at_import=had, hip_initialised_at_import=hip_up, ok=ok, reason=reason)


IPC_MODE = _ipc_mode_guard()

__version__ = '0.1.0'


def set_matrix_products(form):
    """How the fp32 products of the matrix kernels are formed, process-wide (no reference counterpart: torch leaves this to its BLAS).
    'fp32' (default): exact fp32 MFMA products.  'bf16x6': every fp32 operand as three bf16 pieces, each product as its six partial
    products on the bf16 matrix path with fp32 accumulation — the conv / linear rows form (csrc/igemm.hip) and the fused 128-wide
    gradient step on 7x7-grid rows (csrc/ppo_bf16.hpp).  Results agree to the tests' 1e-5, not bit for bit.  Also settable with the
    environment variable PFA_MATRIX_PRODUCTS before the library is first used."""
    from . import _lib
    if form not in ('fp32', 'bf16x6'):
        raise ValueError(f"matrix products {form!r}: expected 'fp32' or 'bf16x6'")
    _lib.check(_lib.lib().pfa_igemm_set_products(1 if form == 'bf16x6' else 0), 'set_matrix_products')


def get_matrix_products():
    from . import _lib
    return 'bf16x6' if _lib.lib().pfa_igemm_get_products() == 1 else 'fp32'
