"""Error types of the boundary, same names and bases as pufferlib/exceptions.py:5-22.

Whenever the reference's ``pufferlib.exceptions`` module is loaded in the process, an error raised here is an instance of
BOTH this module's class and the reference's (a combined subclass, created on first use): a caller of
``pufferlib.vector.make(..., backend=pufferlib_amd.vector.Squared)`` that catches ``pufferlib.exceptions.APIUsageError``
also catches what this backend raises — whichever package was imported first."""
import sys

_combined = {}


def _resolve(cls, name):
    ref_mod = sys.modules.get('pufferlib.exceptions')
    ref = getattr(ref_mod, name, None) if ref_mod is not None else None
    if not isinstance(ref, type) or ref is cls or issubclass(cls, ref):
        return cls
    key = (cls, ref)
    if key not in _combined:
        try:
            _combined[key] = type(name, (cls, ref), {'__module__': cls.__module__})
        except TypeError:            # incompatible layouts: keep our own class
            _combined[key] = cls
    return _combined[key]


class APIUsageError(RuntimeError):
    """Raised when the vecenv / trainer API is used incorrectly."""

    def __new__(cls, *args, **kwargs):
        return RuntimeError.__new__(_resolve(cls, 'APIUsageError') if cls is APIUsageError else cls, *args)

    def __init__(self, message='API usage error.'):
        self.message = message
        RuntimeError.__init__(self, self.message)

    def __reduce__(self):
        # rebuilt through THIS class: the combined subclass _resolve may have produced is not importable by name, and an error
        # that crosses a process boundary (multiprocessing / Ray workers, concurrent.futures) must not become a PicklingError
        return (APIUsageError, (self.message,))


class InvalidAgentError(ValueError):
    def __new__(cls, *args, **kwargs):
        return ValueError.__new__(_resolve(cls, 'InvalidAgentError') if cls is InvalidAgentError else cls, *args)

    def __init__(self, agent_id, agents):
        self.agent_id, self.agents = agent_id, agents
        ValueError.__init__(self, f'Invalid agent/team ({agent_id}) specified. Valid values:\n{agents}')

    def __reduce__(self):
        return (InvalidAgentError, (self.agent_id, self.agents))


class ExtensionError(RuntimeError):
    """The HIP extension is missing or a kernel call failed.  Never swallowed: there is no CPU fallback."""
