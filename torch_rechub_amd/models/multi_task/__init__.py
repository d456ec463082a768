"""Multi-task models over the hot-path layers (names of torch_rechub/models/multi_task/__init__.py)."""
from importlib import import_module

_EXPORTS = {"shared_bottom": "SharedBottom", "esmm": "ESMM", "mmoe": "MMOE", "ple": "PLE", "aitm": "AITM"}
__all__ = []
for _module, _name in _EXPORTS.items():
    globals()[_name] = getattr(import_module(f"{__name__}.{_module}"), _name)
    __all__.append(_name)
del _module, _name
