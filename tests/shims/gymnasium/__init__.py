"""Tests-only stand-in for ``gymnasium`` (see spaces.py).  Provides Env / Wrapper
with the delegation semantics the reference relies on
(pufferlib/postprocess.py:8-23 calls ``super().step``)."""
from . import spaces  # noqa: F401

__version__ = '0.29.1-shim'


class Env:
    metadata = {}
    render_mode = None
    observation_space = None
    action_space = None

    def reset(self, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def render(self):
        return None

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    def reset(self, seed=None, options=None):
        return self.env.reset(seed=seed)

    def step(self, action):
        return self.env.step(action)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()

    @property
    def render_mode(self):
        return self.env.render_mode

    @property
    def unwrapped(self):
        return self.env.unwrapped


def make(*args, **kwargs):
    raise ImportError('gymnasium shim: no registered environments')
