"""Matching models on the hot path (config 5): DSSM (reference torch_rechub/models/matching/dssm.py)."""
from .dssm import DSSM

__all__ = ["DSSM"]
