from . import ranking  # noqa: F401
