"""Multi-task models over the hot-path layers (names of torch_rechub/models/multi_task/__init__.py)."""
from .aitm import AITM
from .esmm import ESMM
from .mmoe import MMOE
from .ple import PLE
from .shared_bottom import SharedBottom

__all__ = ["SharedBottom", "ESMM", "MMOE", "PLE", "AITM"]
