"""Ranking models on the hot path (the names torch_rechub/models/ranking/__init__.py exports, as far as mirrored)."""
from importlib import import_module

# module -> public classes; `from torch_rechub_amd.models.ranking import DeepFM` etc. work as in the reference
_EXPORTS = {
    "afm": ("AFM",),
    "autoint": ("AutoInt",),
    "bst": ("BST",),
    "dcn": ("DCN",),
    "dcn_v2": ("DCNv2",),
    "deepfm": ("DeepFM",),
    "dien": ("DIEN", "AUGRU", "AUGRU_Cell"),
    "din": ("DIN", "ActivationUnit"),
    "edcn": ("EDCN",),
    "fibinet": ("FiBiNet",),
    "widedeep": ("WideDeep",),
}
__all__ = []
for _module, _names in _EXPORTS.items():
    _loaded = import_module(f"{__name__}.{_module}")
    for _n in _names:
        globals()[_n] = getattr(_loaded, _n)
        __all__.append(_n)
del _module, _names, _loaded, _n
