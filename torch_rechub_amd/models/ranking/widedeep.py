"""Wide & Deep (API mirror of torch_rechub/models/ranking/widedeep.py:14-41)."""
import torch

from ...basic.layers import LR, MLP, EmbeddingLayer


class WideDeep(torch.nn.Module):

    def __init__(self, wide_features, deep_features, mlp_params):
        super().__init__()
        self.wide_features = wide_features
        self.deep_features = deep_features
        self.wide_dims = sum(fea.embed_dim for fea in wide_features)
        self.deep_dims = sum(fea.embed_dim for fea in deep_features)
        self.linear = LR(self.wide_dims)
        self.embedding = EmbeddingLayer(wide_features + deep_features)
        self.mlp = MLP(self.deep_dims, **mlp_params)

    def forward(self, x):
        input_wide = self.embedding(x, self.wide_features, squeeze_dim=True)
        input_deep = self.embedding(x, self.deep_features, squeeze_dim=True)
        return self.mlp.sigmoid_head(input_deep, self.linear(input_wide))
