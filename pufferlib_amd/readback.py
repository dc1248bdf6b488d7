"""Device -> host readbacks of the trainer's two per-iteration reports, immediate (default) or deferred.

The reference's ``evaluate`` ends by averaging the episode infos it collected and ``train`` ends by turning the loss
tensors into python floats (clean_pufferl.py:127-152, 249-270).  On the device path both are one small f64 vector copied
into a pinned host buffer behind the kernels that produce it.  By default the host waits for that copy at once, exactly
where the reference has its numbers (two stream syncs per iteration).

``PFA_LAZY_READBACK=1`` defers the wait: the caller gets a *lazy* container back and the first access to the numbers
(``data.stats['score']``, ``data.losses.policy_loss``, ``dict(infos)``, printing, pickling, the dashboard) waits for the
copy's event and fills the container in; a readback is also resolved before its buffer is reused one iteration later, so
the host never runs more than one iteration ahead of the device and an error carried by the numbers (the reset-tape
underrun flag) surfaces at most one ``evaluate()`` late.  It removes the two host round trips per iteration from the
device's timeline.  History of what that is worth on MI355X for the headline workload: nothing in round 2 (351 M deferred
against 356 M immediate), 2-3.5 % on the kernels of round 5's first half (406-416 M against 392-401 M) — and nothing again
since ``evaluate()`` enqueues the update's GAE pass behind its statistics event (clean_pufferl._finish_evaluate: the device
works through the host's round trip) and the waits poll instead of parking the thread (``wait_event`` below): 408.9 / 408.2 M
deferred against 407.5 / 408.2 M immediate on one box (profiles/r05_ab_host_levers.txt).  The mode stays as an option for
hosts with a slow Python side; the GPU test suite passes in both modes.
"""
import os
import time

from .namespace import Namespace


def eager():
    return os.environ.get('PFA_LAZY_READBACK', '0') in ('', '0')


SPIN_WAIT_US = 2000.0      # wait_event's poll window (0 = block at once)


def wait_event(event):
    """Wait for a recorded event the way a latency-bound caller wants it: poll ``event.query()`` for up to SPIN_WAIT_US
    microseconds (2000; 0 = block at once), then fall back to the runtime's blocking wait.  The two readbacks of an
    iteration are waited for while the device still has ~0.2-1 ms of queued work, so the runtime's own wait has long left its
    short active phase and parked the thread by the time the event fires — the wake-up then costs tens of microseconds of idle
    device, twice per iteration.  A polling host thread sees the event within a query's latency.  Long waits (a 0.7 s conv
    update) burn the poll window once and then block as before."""
    spin = SPIN_WAIT_US
    if spin > 0:
        query = event.query
        deadline = time.perf_counter() + spin * 1e-6
        while not query():
            if time.perf_counter() > deadline:
                event.synchronize()
                return
        return
    event.synchronize()


_direct = {'ok': None}


def direct_ok(device):
    """True when a kernel can write the trainer's report numbers straight into pinned host memory: checked ONCE with a real
    launch — pfa_train_log_pack into a pinned buffer, known numbers back after an event wait — so a
    runtime where pinned host memory is not device-writable falls back to the copy instead of reporting garbage."""
    if _direct['ok'] is None:
        ok = True
        if ok:
            try:
                import torch
                from . import _lib
                src = torch.arange(1, 11, dtype=torch.float64, device=device) * 0.5
                host = torch.full((10,), -1.0, dtype=torch.float64).pin_memory()
                _lib.check(_lib.lib().pfa_train_log_pack(_lib.ptr(src[:6]), _lib.ptr(src[6:]), _lib.ptr(host), _lib.stream_handle()), 'log_pack probe')
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                ev.synchronize()
                ok = bool(torch.equal(host, src.cpu()))
            except Exception:
                ok = False
        _direct['ok'] = ok
    return _direct['ok']


class Pending:
    """One in-flight readback: ``submit`` enqueues the copy of ``src`` (device tensor) into this object's pinned buffer on
    the current stream and records the event; ``resolve`` waits for it once and hands the host array to ``finish``."""

    def __init__(self):
        self.host = None
        self.event = None
        self.finish = None

    def submit(self, src, finish, defer=False):
        """``defer``: leave the wait to the caller even in the default (eager) mode — it enqueues more work behind the event first
        and then calls ``resolve()`` itself (clean_pufferl._finish_evaluate: the update's GAE pass runs under the host's wait)."""
        import torch
        self.resolve()                      # the buffer is about to be reused
        if self.host is None or self.host.shape != src.shape or self.host.dtype != src.dtype:
            self.host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            self.event = torch.cuda.Event()
        self.host.copy_(src, non_blocking=True)
        self.event.record(torch.cuda.current_stream(src.device))
        self.finish = finish
        if eager() and not defer:
            self.resolve()
        return self

    def direct_buffer(self, shape, dtype):
        """The pinned host buffer itself, for a kernel that writes its (few) result numbers straight into host memory — no
        device-to-host copy launch behind it.  Any readback still in flight is resolved first (its numbers live in this buffer).
        Follow with ``submit_direct``."""
        import torch
        self.resolve()
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        if self.host is None or tuple(self.host.shape) != shape or self.host.dtype != dtype:
            self.host = torch.empty(shape, dtype=dtype, pin_memory=True)
            self.event = torch.cuda.Event()
        return self.host

    def submit_direct(self, finish, device=None, defer=False):
        """The producing kernel (already enqueued on the current stream) writes into ``direct_buffer``: record the event behind it.
        ``defer``: as in ``submit``."""
        import torch
        self.event.record(torch.cuda.current_stream(device))
        self.finish = finish
        if eager() and not defer:
            self.resolve()
        return self

    @property
    def outstanding(self):
        return self.finish is not None

    def resolve(self):
        finish = self.finish
        if finish is None:
            return
        self.finish = None
        wait_event(self.event)
        finish(self.host.numpy().copy())


class LazyDict(dict):
    """A dict whose contents arrive with a ``Pending`` readback: every read or write first resolves it.  Behaves as a plain
    dict afterwards (and compares, prints, copies and pickles as one).

    Limitation (why lazy mode is opt-in, PFA_LAZY_READBACK=1): C code that reads a dict's storage directly — ``json.dumps``'s C
    encoder, ``PyDict_*`` callers — bypasses the forcing wrappers and sees an unresolved instance as EMPTY.  Hand such consumers
    ``materialize(d)`` (or ``dict(d)``, which forces).  clean_pufferl materialises before it logs to wandb / the dashboard."""
    __slots__ = ('_pending',)

    def __init__(self, pending=None):
        dict.__init__(self)
        self._pending = pending

    def _force(self):
        p = self._pending
        if p is not None:
            self._pending = None
            p.resolve()

    def fill(self, values):
        """Called by the readback's continuation (no forcing: this IS the resolution)."""
        self._pending = None
        dict.update(self, values)

    def __reduce__(self):
        self._force()
        return (dict, (dict(dict.items(self)),))


def materialize(d):
    """A plain dict / namespace with the readback resolved: for consumers that bypass the lazy containers' python-level hooks."""
    if isinstance(d, LazyDict):
        d._force()
        return dict(dict.items(d))
    return d


def _forcing(name):
    base = getattr(dict, name)

    def method(self, *args, **kwargs):
        self._force()
        for other in args:                  # d1 == d2, d1 | d2, d1.update(d2): dict's C code reads the operand's storage directly
            if isinstance(other, LazyDict):
                other._force()
        return base(self, *args, **kwargs)
    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


for _name in ('__getitem__', '__setitem__', '__delitem__', '__iter__', '__len__', '__contains__', '__repr__', '__eq__', '__ne__',
              '__or__', '__ror__', '__ior__', '__reversed__', 'keys', 'values', 'items', 'get', 'copy', 'update', 'pop', 'popitem',
              'setdefault', 'clear'):
    setattr(LazyDict, _name, _forcing(_name))
del _name


class LazyLosses(Namespace):
    """``data.losses`` (clean_pufferl.py:369-378: a namespace of python floats) whose fields arrive with a ``Pending``
    readback.  Attribute, item and iteration access resolve it first."""
    __slots__ = ('_pending',)

    def __init__(self, **fields):
        object.__setattr__(self, '_pending', None)
        Namespace.__init__(self, **fields)

    def attach(self, pending):
        object.__setattr__(self, '_pending', pending)

    def fill(self, **values):
        object.__setattr__(self, '_pending', None)
        object.__getattribute__(self, '__dict__').update(values)

    def __getattribute__(self, name):
        if name == '__dict__' or not name.startswith('_'):
            if name not in ('attach', 'fill'):
                p = object.__getattribute__(self, '_pending')
                if p is not None:
                    object.__setattr__(self, '_pending', None)
                    p.resolve()
        return object.__getattribute__(self, name)

    def __repr__(self):
        return 'namespace(' + ', '.join(f'{k}={v!r}' for k, v in self.__dict__.items()) + ')'
