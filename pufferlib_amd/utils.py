"""The two helpers of pufferlib/utils.py the trainer relies on: a cumulative wall-clock context timer
(utils.py:247-319, elapsed/calls only) and the ``@profile`` decorator that keeps one timer per decorated
function on the first positional argument's ``_timers`` dict (utils.py:321-340; Profile.update reads
``data._timers['evaluate'|'train'].elapsed``, clean_pufferl.py:359,363)."""
import functools
import time


class Profiler:
    def __init__(self):
        self.elapsed = 0.0
        self.calls = 0
        self._prev = 0.0
        self._t0 = None

    @property
    def delta(self):
        d = self.elapsed - self._prev
        self._prev = self.elapsed
        return d

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.elapsed += time.perf_counter() - self._t0
        self.calls += 1

    start = __enter__
    stop = __exit__

    def __repr__(self):
        return f'Elapsed: {self.elapsed:.4f} s, Calls: {self.calls}'


def profile(func):
    name = func.__name__

    @functools.wraps(func)
    def wrapper(owner, *args, **kwargs):
        timers = owner.__dict__.setdefault('_timers', {})
        timer = timers.get(name)
        if timer is None:
            timer = timers[name] = Profiler()
        with timer:
            return func(owner, *args, **kwargs)

    return wrapper


def unroll_nested_dict(d):
    """Yield (flattened_key, leaf) pairs, '/'-joined (pufferlib/utils.py:56-65)."""
    if not isinstance(d, dict):
        return d
    for k, v in d.items():
        if isinstance(v, dict):
            for k2, v2 in unroll_nested_dict(v):
                yield f'{k}/{k2}', v2
        else:
            yield k, v
