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
from .namespace import namespace, Namespace  # noqa: F401

__version__ = '0.1.0'
