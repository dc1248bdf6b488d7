"""Base classes live here; ``pettingzoo/__init__`` re-exports them."""


class ParallelEnv:
    metadata = {}

    def reset(self, seed=None, options=None):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def close(self):
        pass


class AECEnv:
    metadata = {}
