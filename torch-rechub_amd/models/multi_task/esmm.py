"""Entire Space Multi-task Model (API mirror of torch_rechub/models/multi_task/esmm.py:13-56).

User and item embeddings are concatenated (the reference gathers the user list twice, :37-38; once here), two towers
give pCVR and pCTR, and the output columns are [pCVR, pCTR, pCTR * pCVR]; the trainer sums the losses of columns 1 and 2
only (mtl_trainer.py:122-124).  ``tower_dims`` keeps the reference's arithmetic: every feature is assumed to have the
embed_dim of the first feature of its list."""
import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer


class ESMM(nn.Module):

    def __init__(self, user_features, item_features, cvr_params, ctr_params):
        super().__init__()
        self.user_features, self.item_features = user_features, item_features
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.tower_dims = (len(user_features) * user_features[0].embed_dim +
                           len(item_features) * item_features[0].embed_dim)
        self.tower_cvr = MLP(self.tower_dims, **cvr_params)
        self.tower_ctr = MLP(self.tower_dims, **ctr_params)

    def forward(self, x):
        both = self.embedding(x, self.user_features + self.item_features, squeeze_dim=False)  # user fields, then item
        tower_in = both.flatten(start_dim=1)
        cvr = torch.sigmoid(self.tower_cvr(tower_in))
        ctr = torch.sigmoid(self.tower_ctr(tower_in))
        return torch.cat([cvr, ctr, ctr * cvr], dim=1)
