"""Ranking models on the hot path (reference torch_rechub/models/ranking/__init__.py)."""
from .afm import AFM
from .autoint import AutoInt
from .dcn import DCN
from .dcn_v2 import DCNv2
from .deepfm import DeepFM
from .din import DIN, ActivationUnit
from .edcn import EDCN
from .fibinet import FiBiNet
from .widedeep import WideDeep

__all__ = ["WideDeep", "DeepFM", "DCN", "DCNv2", "DIN", "AFM", "FiBiNet", "AutoInt", "EDCN"]
