"""Deep & Cross Network (API mirror of torch_rechub/models/ranking/dcn.py:14-38).

Reference forward: flattened gather -> CrossNetwork and MLP (no output layer) side by side -> LR on their
concatenation -> sigmoid.  Attribute names / checkpoint keys are the reference's (``embedding``, ``cn``, ``mlp``,
``linear.fc``).  Here the gather is one fused HIP launch, the cross stack one launch per direction
(``rh_cross_fwd/bwd``), the MLP the fused BatchNorm path, and the final LR + sigmoid the head kernel when its width
allows (up to 1024 columns).
"""
import torch
from torch import nn

from ... import ops
from ...basic.layers import LR, MLP, CrossNetwork, EmbeddingLayer


class DCN(nn.Module):

    def __init__(self, features, n_cross_layers, mlp_params):
        super().__init__()
        width = sum(f.embed_dim for f in features)
        self.features, self.dims = features, width
        self.embedding = EmbeddingLayer(features)
        self.cn = CrossNetwork(width, n_cross_layers)
        self.mlp = MLP(width, output_layer=False, **mlp_params)
        self.linear = LR(width + mlp_params["dims"][-1])

    def _head(self, z):
        fc = self.linear.fc
        if ops.head_ok(z, fc, ()):
            return ops.head_sigmoid(z, fc.weight, fc.bias)
        return torch.sigmoid(fc(z).squeeze(1))

    def forward(self, x):
        h = self.embedding(x, self.features, squeeze_dim=True)
        return self._head(torch.cat((self.cn(h), self.mlp(h)), dim=1))
