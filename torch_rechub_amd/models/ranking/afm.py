"""AFM (API mirror of torch_rechub/models/ranking/afm.py:16-65): first-order LR + attention-weighted FM vector.

Reference forward: gather -> LR on the flattened embeddings -> FM(reduce_sum=False) (the (B, D) second-order
vector) -> attention(relu(Linear(D, t)) @ h) -> softmax over dim 1 -> (att * fm) @ p -> sigmoid.  The attention
logit has shape (B, 1) and the softmax runs over that size-1 dimension (afm.py:51), so the attention weight is
identically 1 and ``attention_liner`` / ``h`` receive zero gradient: reproduced as is, parameter names included
(``attention_liner``, ``h``, ``p``).  The gather + LR run as the fused HIP launch, the FM vector as ``rh_fm_fwd``.
"""
import torch
from torch import nn

from ... import ops
from ...basic.layers import FM, LR, EmbeddingLayer


class AFM(nn.Module):

    def __init__(self, fm_features, embed_dim, t=64):
        super().__init__()
        self.fm_features = fm_features
        self.embed_dim = embed_dim
        self.fm_dims = sum(fea.embed_dim for fea in fm_features)
        self.linear = LR(self.fm_dims)
        self.fm = FM(reduce_sum=False)
        self.embedding = EmbeddingLayer(fm_features)
        self.attention_liner = nn.Linear(self.embed_dim, t)
        self.h = nn.init.xavier_uniform_(nn.Parameter(torch.empty(t, 1)))
        self.p = nn.init.xavier_uniform_(nn.Parameter(torch.empty(self.embed_dim, 1)))

    def attention(self, y_fm):
        score = torch.relu(self.attention_liner(y_fm)) @ self.h  # (B, 1)
        return torch.softmax(score, dim=1)  # over the size-1 dimension, as the reference does

    def forward(self, x):
        emb = self.embedding
        if emb.can_fuse(x, self.fm_features):
            flat, _, y_linear = emb.fused(x, self.fm_features, (), self.linear.fc.weight, self.linear.fc.bias)
            input_fm = flat.reshape(flat.shape[0], len(self.fm_features), -1)
        else:
            input_fm = emb(x, self.fm_features, squeeze_dim=False)
            y_linear = self.linear(input_fm.flatten(start_dim=1))
        y_fm = self.fm(input_fm)
        outs = (self.attention(y_fm) * y_fm) @ self.p
        return torch.sigmoid((y_linear + outs).squeeze(1))
