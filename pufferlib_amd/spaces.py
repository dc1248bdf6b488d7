"""Minimal observation/action space descriptors (shape, dtype, n, contains) so the package does not depend
on gymnasium at run time.  They carry exactly what the hot path reads from
``single_observation_space`` / ``single_action_space`` (clean_pufferl.py:40-43, models.py:26-37,
vector.py:36-39,55-68).  If gymnasium is installed its spaces work as well (duck-typed)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.can_cast(x.dtype, self.dtype)
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    def __repr__(self):
        return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == () and np.issubdtype(x.dtype, np.integer) and 0 <= int(x) < self.n)

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n

    def __repr__(self):
        return f'Discrete({self.n})'


class MultiDiscrete:
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.asarray(nvec, dtype=dtype)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.issubdtype(x.dtype, np.integer)
                    and np.all(x >= 0) and np.all(x < self.nvec))

    def __len__(self):
        return len(self.nvec)

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)

    def __repr__(self):
        return f'MultiDiscrete({self.nvec})'
