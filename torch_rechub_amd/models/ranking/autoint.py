"""AutoInt (API mirror of torch_rechub/models/ranking/autoint.py:14-102): multi-head self-attention over the fields.

Reference forward: sparse gather (B, Fs, D) + one Linear(1, D, bias=False) per dense feature -> (B, F, D) ->
InteractingLayer x num_layers -> Linear(F*D, 1) on the attention output + LR on the raw embeddings (+ MLP) -> sigmoid.
Same constructor, attribute names and state_dict keys (``sparse_embedding``, ``dense_embeddings.<name>``,
``interacting_layers.{l}.W_{Q,K,V,Res}``, ``linear.fc``, ``attn_linear``, ``mlp``).  The gather is the fused HIP launch;
the four projections of an interacting layer run as ONE GEMM over the concatenated weights.
"""
import torch
from torch import nn

from ... import ops
from ...basic.layers import LR, MLP, EmbeddingLayer, InteractingLayer


class AutoInt(nn.Module):

    def __init__(self, sparse_features, dense_features, num_layers=3, num_heads=2, dropout=0.0, mlp_params=None):
        super().__init__()
        self.sparse_features = sparse_features
        self.dense_features = dense_features if dense_features is not None else []
        if len(self.sparse_features) == 0:
            raise ValueError("AutoInt requires at least one sparse feature to determine embed_dim.")
        self.embed_dim = self.sparse_features[0].embed_dim
        self.num_sparse = len(self.sparse_features)
        self.num_dense = len(self.dense_features)
        self.num_fields = self.num_sparse + self.num_dense
        self.dims = self.num_fields * self.embed_dim
        self.num_layers = num_layers
        self.sparse_embedding = EmbeddingLayer(self.sparse_features)
        self.dense_embeddings = nn.ModuleDict({fea.name: nn.Linear(1, self.embed_dim, bias=False)
                                               for fea in self.dense_features})
        self.interacting_layers = nn.ModuleList([InteractingLayer(self.embed_dim, num_heads=num_heads, dropout=dropout,
                                                                  residual=True) for _ in range(num_layers)])
        self.linear = LR(self.dims)
        self.attn_linear = nn.Linear(self.dims, 1)
        self.use_mlp = mlp_params is not None
        if self.use_mlp:
            self.mlp = MLP(self.dims, **mlp_params)

    def forward(self, x):
        emb = self.sparse_embedding
        if emb.can_fuse(x, self.sparse_features):
            flat, _, _ = emb.fused(x, self.sparse_features, ())
            sparse_emb = flat.reshape(flat.shape[0], self.num_sparse, self.embed_dim)
        else:
            sparse_emb = emb(x, self.sparse_features, squeeze_dim=False)
        if self.dense_features:
            # Linear(1, D, bias=False) on a scalar is value * weight column: all dense fields in one broadcast
            vals = torch.stack([x[f.name].float().reshape(-1) for f in self.dense_features], dim=1)  # (B, Fd)
            cols = torch.stack([self.dense_embeddings[f.name].weight[:, 0] for f in self.dense_features])  # (Fd, D)
            embed_x = torch.cat([sparse_emb, vals.unsqueeze(-1) * cols.unsqueeze(0)], dim=1)
        else:
            embed_x = sparse_emb
        embed_flat = embed_x.flatten(start_dim=1)
        attn_out = embed_x
        for layer in self.interacting_layers:
            attn_out = layer(attn_out)
        y = self.attn_linear(attn_out.flatten(start_dim=1)) + self.linear(embed_flat)
        if self.use_mlp:
            return self.mlp.sigmoid_head(embed_flat, y)
        return torch.sigmoid(y.squeeze(1))
