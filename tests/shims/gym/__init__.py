"""Tests-only stand-in for legacy ``gym``: re-exports the gymnasium shim so that
``pufferlib/spaces.py:1-9`` can build its (gym, gymnasium) isinstance tuples."""
from gymnasium import Env, Wrapper, spaces, make  # noqa: F401

__version__ = '0.23.0-shim'
