"""Embedding-table initializers (API mirror of torch_rechub/basic/initializers.py:4-100).

Every initializer is a callable ``(vocab_size, embed_dim, padding_idx=None) -> nn.Embedding`` and
zeroes the ``padding_idx`` row after filling the table, exactly like the reference.  The tables stay
plain ``nn.Embedding`` modules (RegularizationLoss and the checkpoint keys depend on that); the HIP
kernels read ``embedding.weight`` in place through a pointer table.
"""
import torch
from torch import nn


KERNEL_DIMS = (4, 8, 16, 32, 64, 128)  # row widths the gather / optimizer kernels handle (16-byte lanes, 2^k per row)


def padded_dim(embed_dim):
    """Physical row width of a table of logical width ``embed_dim``: the next kernel width (None beyond 128)."""
    for d in KERNEL_DIMS:
        if embed_dim <= d:
            return d
    return None


class PaddedEmbedding(nn.Embedding):
    """``nn.Embedding`` of a width the kernels do not address directly (the reference's automatic ``embed_dim`` =
    floor(6 V^0.25) is routinely 10, 18, 33 ...; features.py:54-60, utils/data.py:86-101): the rows are STORED padded to
    the next kernel width (10 -> 16, 18 -> 32 floats) with zeros that stay zero -- no gradient ever reaches them (the
    layers cut the padding columns off before anything consumes the rows) and Adam / weight decay map 0 to 0.

    Checkpoint ABI unchanged: ``state_dict()`` exposes ``weight`` as the (vocab, embed_dim) view of the storage and
    ``load_state_dict`` accepts (vocab, embed_dim) -- a reference ``model.pth`` loads, and what this writes loads there.
    ``embedding_dim`` is the logical width; ``weight`` is the physical (vocab, padded) parameter the kernels read."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None):
        phys = padded_dim(embedding_dim)
        if phys is None:
            raise ValueError(f"embed_dim {embedding_dim} > 128 has no HIP gather kernel")
        super().__init__(num_embeddings, phys, padding_idx=padding_idx)
        self.logical_dim = int(embedding_dim)
        self.embedding_dim = int(embedding_dim)
        with torch.no_grad():
            self.weight[:, self.logical_dim:].zero_()
        self.weight._rh_logical_dim = self.logical_dim  # optim.TableAdam slices / pads this table's moments in checkpoints
        self._register_state_dict_hook(PaddedEmbedding._slice_hook)
        self._register_load_state_dict_pre_hook(self._pad_hook)

    @staticmethod
    def _slice_hook(module, state_dict, prefix, local_metadata):
        key = prefix + "weight"
        if key in state_dict:
            state_dict[key] = state_dict[key][:, :module.logical_dim]

    def _pad_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + "weight"
        w = state_dict.get(key)
        if w is not None and w.dim() == 2 and w.shape[1] == self.logical_dim and self.logical_dim != self.weight.shape[1]:
            full = w.new_zeros((w.shape[0], self.weight.shape[1]))
            full[:, :self.logical_dim] = w
            state_dict[key] = full

    def forward(self, input):  # direct (eager) use: logical rows
        return torch.nn.functional.embedding(input, self.weight, self.padding_idx)[..., :self.logical_dim]

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.logical_dim} (stored {self.weight.shape[1]} wide)"


def make_table(vocab_size, embed_dim, padding_idx, fill):
    """nn.Embedding for a kernel width, PaddedEmbedding otherwise; ``fill(weight_view)`` sees the LOGICAL (vocab, D) view."""
    if embed_dim in KERNEL_DIMS or padded_dim(embed_dim) is None:
        table = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx)
        fill(table.weight)
        return table
    table = PaddedEmbedding(vocab_size, embed_dim, padding_idx=padding_idx)
    tmp = torch.empty((vocab_size, embed_dim), dtype=table.weight.dtype, device=table.weight.device)
    fill(tmp)
    with torch.no_grad():
        table.weight[:, :embed_dim] = tmp
    return table


def _finish(table, padding_idx):
    if padding_idx is not None:
        with torch.no_grad():
            table.weight[padding_idx].zero_()
    return table


class _Filler(object):
    """Base: build the nn.Embedding, delegate the fill to ``_fill(weight)``."""

    def __call__(self, vocab_size, embed_dim, padding_idx=None):
        table = make_table(vocab_size, embed_dim, padding_idx, lambda w: self._fill(w.detach()))
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
        if embed_dim in KERNEL_DIMS or padded_dim(embed_dim) is None:
            return nn.Embedding.from_pretrained(self.embedding_weight, freeze=self.freeze, padding_idx=padding_idx)
        table = make_table(vocab_size, embed_dim, padding_idx, lambda w: w.copy_(self.embedding_weight))
        table.weight.requires_grad_(not self.freeze)
        return table
