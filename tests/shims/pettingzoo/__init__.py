"""Tests-only stand-in for ``pettingzoo`` base classes (ocean.py:149 subclasses
ParallelEnv at import time)."""
from . import utils  # noqa: F401
from .utils.env import ParallelEnv, AECEnv  # noqa: F401
