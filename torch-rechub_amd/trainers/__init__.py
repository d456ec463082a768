from .ctr_trainer import CTRTrainer  # noqa: F401
