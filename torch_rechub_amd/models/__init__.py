from . import matching, ranking  # noqa: F401
