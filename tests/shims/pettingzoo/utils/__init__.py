from . import env  # noqa: F401
