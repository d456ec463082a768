"""RegularizationLoss (API mirror of torch_rechub/basic/loss_func.py:6-68).

Same classification rules: parameters of normalisation layers are skipped, parameters of
``nn.Embedding`` / ``nn.EmbeddingBag`` modules use the embedding coefficients, everything else the
dense ones.  Returns the python float 0.0 when nothing applies (the trainer adds it to the loss).
"""
from torch import nn

_NORMS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.LayerNorm, nn.GroupNorm, nn.InstanceNorm1d,
          nn.InstanceNorm2d, nn.InstanceNorm3d)


class RegularizationLoss(nn.Module):

    def __init__(self, embedding_l1=0.0, embedding_l2=0.0, dense_l1=0.0, dense_l2=0.0):
        super().__init__()
        self.embedding_l1 = embedding_l1
        self.embedding_l2 = embedding_l2
        self.dense_l1 = dense_l1
        self.dense_l2 = dense_l2

    def active(self):
        return max(self.embedding_l1, self.embedding_l2, self.dense_l1, self.dense_l2) > 0

    def embedding_term(self, model):
        """Only the nn.Embedding part of the penalty (python 0.0 when it is off).  The data-parallel trainers add
        (1 - 1/world) of it back after scaling the loss by 1/world: a table's regulariser gradient is a local dense
        term that takes no part in the row exchange, so it must not be divided by the world size."""
        if max(self.embedding_l1, self.embedding_l2) <= 0:
            return 0.0
        return self.forward(model, _only_tables=True)

    @staticmethod
    def _exchanged_tables(model):
        """ids of the embedding parameters whose gradient travels by the row exchange (not by the dense all-reduce): every
        nn.Embedding except those a model flags ``_rh_dense`` (a small table read as a slice, BST's positional table) --
        the same rule as distributed.table_parameters(), so a dense-bucket embedding is not added back world times."""
        ids = set()
        for m in model.modules():
            if isinstance(m, (nn.Embedding, nn.EmbeddingBag)) and not getattr(m, "_rh_dense", False):
                ids.update(id(p) for p in m.parameters())
        return ids

    def forward(self, model, _only_tables=False):
        total = 0.0
        if not self.active():
            return total
        skip, tables = set(), set()
        exchanged = self._exchanged_tables(model) if _only_tables else ()
        for m in model.modules():
            if isinstance(m, _NORMS):
                skip.update(id(p) for p in m.parameters())
            elif isinstance(m, (nn.Embedding, nn.EmbeddingBag)):
                tables.update(id(p) for p in m.parameters())
        for p in model.parameters():
            if not p.requires_grad or id(p) in skip:
                continue
            if _only_tables and id(p) not in exchanged:
                continue
            l1, l2 = (self.embedding_l1, self.embedding_l2) if id(p) in tables else (self.dense_l1, self.dense_l2)
            if l1 > 0:
                total = total + l1 * p.abs().sum()
            if l2 > 0:
                total = total + l2 * (p * p).sum()
        return total


class BPRLoss(nn.Module):
    """-log(sigmoid(pos - neg)).mean() (API mirror of torch_rechub/basic/loss_func.py:95-107)."""

    def forward(self, pos_score, neg_score, in_batch_neg=False):
        pos_score = pos_score.view(-1)
        diff = pos_score - neg_score if neg_score.dim() == 1 else pos_score.view(-1, 1) - neg_score
        return -diff.sigmoid().log().mean()
