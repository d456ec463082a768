from .ctr_trainer import CTRTrainer  # noqa: F401
from .match_trainer import MatchTrainer  # noqa: F401
from .mtl_trainer import MTLTrainer  # noqa: F401
