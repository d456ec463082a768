"""Embedding-table initializers (API mirror of torch_rechub/basic/initializers.py:4-100).

Every initializer is a callable ``(vocab_size, embed_dim, padding_idx=None) -> nn.Embedding`` and
zeroes the ``padding_idx`` row after filling the table, exactly like the reference.  The tables stay
plain ``nn.Embedding`` modules (RegularizationLoss and the checkpoint keys depend on that); the HIP
kernels read ``embedding.weight`` in place through a pointer table.
"""
import torch
from torch import nn


def _finish(table, padding_idx):
    if padding_idx is not None:
        with torch.no_grad():
            table.weight[padding_idx].zero_()
    return table


class _Filler(object):
    """Base: build the nn.Embedding, delegate the fill to ``_fill(weight)``."""

    def __call__(self, vocab_size, embed_dim, padding_idx=None):
        table = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx)
        self._fill(table.weight)
        return _finish(table, padding_idx)


class RandomNormal(_Filler):
    """N(mean, std) fill (reference initializers.py:4-21)."""

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = mean, std

    def _fill(self, w):
        nn.init.normal_(w, self.mean, self.std)


class RandomUniform(_Filler):
    """U(minval, maxval) fill (reference initializers.py:24-41)."""

    def __init__(self, minval=0.0, maxval=1.0):
        self.minval, self.maxval = minval, maxval

    def _fill(self, w):
        nn.init.uniform_(w, self.minval, self.maxval)


class XavierNormal(_Filler):
    """Glorot normal fill (reference initializers.py:44-61)."""

    def __init__(self, gain=1.0):
        self.gain = gain

    def _fill(self, w):
        nn.init.xavier_normal_(w, self.gain)


class XavierUniform(_Filler):
    """Glorot uniform fill (reference initializers.py:64-81)."""

    def __init__(self, gain=1.0):
        self.gain = gain

    def _fill(self, w):
        nn.init.xavier_uniform_(w, self.gain)


class Pretrained(object):
    """Table from a given 2-D weight, frozen by default (reference initializers.py:84-100)."""

    def __init__(self, embedding_weight, freeze=True):
        self.embedding_weight = torch.as_tensor(embedding_weight, dtype=torch.float32).clone()
        self.freeze = freeze

    def __call__(self, vocab_size, embed_dim, padding_idx=None):
        rows, cols = self.embedding_weight.shape
        assert vocab_size == rows and embed_dim == cols
        return nn.Embedding.from_pretrained(self.embedding_weight, freeze=self.freeze, padding_idx=padding_idx)
