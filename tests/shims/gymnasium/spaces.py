"""Tests-only stand-in for ``gymnasium.spaces`` (the real package is not installed
and there is no network).  Only what the *unmodified* reference under
/root/reference needs in order to import and to run ocean envs through
``pufferlib.vector.Serial``:  Box / Discrete / MultiDiscrete / MultiBinary / Dict /
Tuple with ``shape``, ``dtype``, ``contains`` and equality.  Semantics that affect
results follow gymnasium 0.29: Box default dtype float32, Discrete/MultiDiscrete
dtype int64, Dict keeps sorted keys.  Never imported by the product package.
"""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)

    @property
    def shape(self):
        return self._shape

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        shape = tuple(int(s) for s in shape)
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()

    def contains(self, x):
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        rng = getattr(self, '_rng', np.random)
        if np.issubdtype(self.dtype, np.floating):
            return rng.uniform(self.low, self.high, self.shape).astype(self.dtype)
        return rng.integers(self.low, self.high + 1, self.shape).astype(self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and self.dtype == other.dtype
                and np.array_equal(self.low, other.low)
                and np.array_equal(self.high, other.high))

    def __repr__(self):
        return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)

    def contains(self, x):
        if isinstance(x, (int, np.integer)):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return self.start <= v < self.start + self.n

    def sample(self):
        return int(np.random.randint(self.n)) + self.start

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start

    def __repr__(self):
        return f'Discrete({self.n})'


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype)

    def contains(self, x):
        if isinstance(x, (list, tuple)):
            x = np.asarray(x)
        return bool(isinstance(x, np.ndarray) and x.shape == self.shape
                    and x.dtype != object and np.issubdtype(x.dtype, np.integer)
                    and np.all(x >= 0) and np.all(x < self.nvec))

    def sample(self):
        return (np.random.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def __len__(self):
        return len(self.nvec)

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)

    def __repr__(self):
        return f'MultiDiscrete({self.nvec})'


class MultiBinary(Space):
    def __init__(self, n):
        self.n = n
        shape = (n,) if np.isscalar(n) else tuple(n)
        super().__init__(shape, np.int8)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all((x == 0) | (x == 1)))

    def __eq__(self, other):
        return isinstance(other, MultiBinary) and self.shape == other.shape


class Dict(Space):
    def __init__(self, spaces=None, **kwargs):
        super().__init__(None, None)
        spaces = dict(spaces or {}, **kwargs)
        self.spaces = {k: spaces[k] for k in sorted(spaces)}

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def values(self):
        return self.spaces.values()

    def items(self):
        return self.spaces.items()

    def contains(self, x):
        return (isinstance(x, dict) and x.keys() == self.spaces.keys()
                and all(self.spaces[k].contains(x[k]) for k in self.spaces))

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __eq__(self, other):
        return isinstance(other, Dict) and self.spaces == other.spaces


class Tuple(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def contains(self, x):
        return (isinstance(x, (tuple, list)) and len(x) == len(self.spaces)
                and all(s.contains(v) for s, v in zip(self.spaces, x)))

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces
