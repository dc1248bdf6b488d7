"""Dict-like attribute bag, same behaviour as pufferlib.namespace (pufferlib/namespace.py:23-55):
``namespace(a=1)`` builds one; instances support attribute access, ``ns['a']``, iteration over keys,
``len``, ``keys/values/items``."""
from collections.abc import Mapping
from types import SimpleNamespace


class Namespace(SimpleNamespace, Mapping):
    def __getitem__(self, key):
        return self.__dict__[key]

    def __iter__(self):
        return iter(self.__dict__)

    def __len__(self):
        return len(self.__dict__)

    def keys(self):
        return self.__dict__.keys()

    def values(self):
        return self.__dict__.values()

    def items(self):
        return self.__dict__.items()


def namespace(self=None, **kwargs):
    if self is None:
        return Namespace(**kwargs)
    self.__dict__.update(kwargs)
