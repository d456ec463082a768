"""Feature descriptors (API mirror of torch_rechub/basic/features.py:5-87).

``SparseFeature`` / ``SequenceFeature`` own and CACHE their ``nn.Embedding`` on first use
(reference features.py:36-39, 68-71), so two models built from the same feature objects share
tables.  The kernels honour that: tables are addressed through per-field pointers,
never copied into a private packed buffer.
"""
import numpy as np

from .initializers import RandomNormal


def get_auto_embedding_dim(num_classes):
    """floor(6 * num_classes ** 0.25), the DCN rule used by the reference (utils/data.py:86-101)."""
    return int(np.floor(6 * np.power(num_classes, 0.25)))


class _TableFeature(object):

    def get_embedding_layer(self):
        if not hasattr(self, "embed"):
            self.embed = self.initializer(self.vocab_size, self.embed_dim, padding_idx=self.padding_idx)
        return self.embed


class SequenceFeature(_TableFeature):
    """Behaviour-sequence / multi-hot feature; values must be padded to a fixed length.

    Args mirror reference features.py:5-31: ``pooling`` in {"mean", "sum", "concat"}.
    """

    def __init__(self, name, vocab_size, embed_dim=None, pooling="mean", shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001)):
        self.name = name
        self.vocab_size = vocab_size
        self.embed_dim = get_auto_embedding_dim(vocab_size) if embed_dim is None else embed_dim
        self.pooling = pooling
        self.shared_with = shared_with
        self.padding_idx = padding_idx
        self.initializer = initializer

    def __repr__(self):
        return f"<SequenceFeature {self.name} with Embedding shape ({self.vocab_size}, {self.embed_dim})>"


class SparseFeature(_TableFeature):
    """Categorical feature (reference features.py:42-71)."""

    def __init__(self, name, vocab_size, embed_dim=None, shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001)):
        self.name = name
        self.vocab_size = vocab_size
        self.embed_dim = get_auto_embedding_dim(vocab_size) if embed_dim is None else embed_dim
        self.shared_with = shared_with
        self.padding_idx = padding_idx
        self.initializer = initializer

    def __repr__(self):
        return f"<SparseFeature {self.name} with Embedding shape ({self.vocab_size}, {self.embed_dim})>"


class DenseFeature(object):
    """Numeric feature, ``embed_dim`` = its width (reference features.py:74-87)."""

    def __init__(self, name, embed_dim=1):
        self.name = name
        self.embed_dim = embed_dim

    def __repr__(self):
        return f"<DenseFeature {self.name}>"
