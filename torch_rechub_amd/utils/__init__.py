from . import data, match  # noqa: F401
