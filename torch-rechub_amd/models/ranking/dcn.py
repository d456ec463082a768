"""Deep & Cross Network (API mirror of torch_rechub/models/ranking/dcn.py:14-38)."""
import torch

from ...basic.layers import LR, MLP, CrossNetwork, EmbeddingLayer


class DCN(torch.nn.Module):

    def __init__(self, features, n_cross_layers, mlp_params):
        super().__init__()
        self.features = features
        self.dims = sum(fea.embed_dim for fea in features)
        self.embedding = EmbeddingLayer(features)
        self.cn = CrossNetwork(self.dims, n_cross_layers)
        self.mlp = MLP(self.dims, output_layer=False, **mlp_params)
        self.linear = LR(self.dims + mlp_params["dims"][-1])

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        cn_out = self.cn(embed_x)
        mlp_out = self.mlp(embed_x)
        y = self.linear(torch.cat([cn_out, mlp_out], dim=1))
        return torch.sigmoid(y.squeeze(1))
