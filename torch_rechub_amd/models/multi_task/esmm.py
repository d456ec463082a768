"""Entire Space Multi-task Model (API mirror of torch_rechub/models/multi_task/esmm.py:13-56).

User and item embeddings are concatenated (the reference gathers the user list twice, :37-38; once here), two towers
give pCVR and pCTR, and the output columns are [pCVR, pCTR, pCTR * pCVR]; the trainer sums the losses of columns 1 and 2
only (mtl_trainer.py:122-124).  ``tower_dims`` keeps the reference's arithmetic: every feature is assumed to have the
embed_dim of the first feature of its list."""
import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer


def _uniform_width(features):
    """Input width the reference assumes for a feature list: len(list) x embed_dim of its first member (esmm.py:27)."""
    return len(features) * features[0].embed_dim


class ESMM(nn.Module):

    def __init__(self, user_features, item_features, cvr_params, ctr_params):
        super().__init__()
        self.user_features, self.item_features = user_features, item_features
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.tower_dims = _uniform_width(user_features) + _uniform_width(item_features)
        for name, params in (("tower_cvr", cvr_params), ("tower_ctr", ctr_params)):  # registration order = key order
            setattr(self, name, MLP(self.tower_dims, **params))

    def forward(self, x):
        fields = self.embedding(x, self.user_features + self.item_features, squeeze_dim=False)  # user fields, then item
        tower_in = fields.flatten(start_dim=1)
        p_cvr, p_ctr = (torch.sigmoid(tower(tower_in)) for tower in (self.tower_cvr, self.tower_ctr))
        return torch.cat([p_cvr, p_ctr, p_ctr * p_cvr], dim=1)
