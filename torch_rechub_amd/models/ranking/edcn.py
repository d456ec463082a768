"""EDCN (API mirror of torch_rechub/models/ranking/edcn.py:15-101): cross and deep streams coupled layer by layer.

Reference forward: flattened gather -> RegulationModule (per-field gates) -> per layer: cross_i += CrossLayer(cross_0,
cross_i); deep_i = MLP_i(deep_i); bridge_i = Bridge(cross_i, deep_i); next layer regulates bridge_i -> LR on
[cross, deep, bridge] -> sigmoid.  Same constructor (it overwrites ``mlp_params["dims"]`` in place, as the reference
does), attribute names and state_dict keys (``cross_layers.{l}.{w,b}``, ``bridge_modules``, ``regulation_modules.{l}.
{g1,g2}``, ``mlps.{l}.mlp.*``, ``linear.fc``).  Quirk kept: the regulation gate is ``softmax`` of ONE scalar per field
(edcn.py:94-95), i.e. identically 1 with zero gradient to g1 / g2.  The gather is the fused HIP launch, the MLPs the fused
BatchNorm path.
"""
import torch
from torch import nn

from ...basic.layers import LR, MLP, CrossLayer, EmbeddingLayer


class EDCN(nn.Module):

    def __init__(self, features, n_cross_layers, mlp_params, bridge_type="hadamard_product", use_regulation_module=True,
                 temperature=1):
        super().__init__()
        self.features = features
        self.n_cross_layers = n_cross_layers
        self.num_fields = len(features)
        self.dims = sum(fea.embed_dim for fea in features)
        self.fea_dims = [fea.embed_dim for fea in features]
        self.embedding = EmbeddingLayer(features)
        self.cross_layers = nn.ModuleList([CrossLayer(self.dims) for _ in range(n_cross_layers)])
        self.bridge_modules = nn.ModuleList([BridgeModule(self.dims, bridge_type) for _ in range(n_cross_layers)])
        self.regulation_modules = nn.ModuleList([
            RegulationModule(self.num_fields, self.fea_dims, tau=temperature, use_regulation=use_regulation_module)
            for _ in range(n_cross_layers)
        ])
        mlp_params["dims"] = [self.dims, self.dims]
        self.mlps = nn.ModuleList([MLP(self.dims, output_layer=False, **mlp_params) for _ in range(n_cross_layers)])
        self.linear = LR(self.dims * 3)

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        cross_i, deep_i = self.regulation_modules[0](embed_x)
        cross_0 = cross_i
        bridge_i = None
        for i in range(self.n_cross_layers):
            if i > 0:
                cross_i, deep_i = self.regulation_modules[i](bridge_i)
            cross_i = cross_i + self.cross_layers[i](cross_0, cross_i)
            deep_i = self.mlps[i](deep_i)
            bridge_i = self.bridge_modules[i](cross_i, deep_i)
        y = self.linear(torch.cat([cross_i, deep_i, bridge_i], dim=1))
        return torch.sigmoid(y.squeeze(1))


class BridgeModule(nn.Module):
    """How the two streams exchange information after each layer (edcn.py:58-79)."""

    KINDS = ("hadamard_product", "pointwise_addition", "concatenation", "attention_pooling")

    def __init__(self, input_dim, bridge_type):
        super().__init__()
        assert bridge_type in self.KINDS, f"bridge_type={bridge_type} is not supported"
        self.bridge_type = bridge_type
        if bridge_type == "concatenation":
            self.concat_pooling = nn.Sequential(nn.Linear(input_dim * 2, input_dim), nn.ReLU())
        elif bridge_type == "attention_pooling":

            def gate():
                return nn.Sequential(nn.Linear(input_dim, input_dim), nn.ReLU(), nn.Linear(input_dim, input_dim, bias=False),
                                     nn.Softmax(dim=-1))

            self.attention_x, self.attention_h = gate(), gate()

    def forward(self, x, h):
        if self.bridge_type == "hadamard_product":
            return x * h
        if self.bridge_type == "pointwise_addition":
            return x + h
        if self.bridge_type == "concatenation":
            return self.concat_pooling(torch.cat([x, h], dim=-1))
        return self.attention_x(x) * x + self.attention_h(h) * h


class RegulationModule(nn.Module):
    """Field-wise gates for the two streams (edcn.py:82-101): softmax of a single scalar per field, hence == 1."""

    def __init__(self, num_fields, dims, tau, use_regulation=True):
        super().__init__()
        self.use_regulation = use_regulation
        if use_regulation:
            self.num_fields, self.dims, self.tau = num_fields, dims, tau
            self.g1 = nn.Parameter(torch.ones(num_fields))
            self.g2 = nn.Parameter(torch.ones(num_fields))

    def _gate(self, g):
        per_field = torch.stack([(g[i] / self.tau).softmax(dim=-1) for i in range(self.num_fields)])  # (F,), all ones
        return torch.repeat_interleave(per_field, torch.tensor(self.dims, device=g.device)).unsqueeze(0)

    def forward(self, x):
        if not self.use_regulation:
            return x, x
        return self._gate(self.g1) * x, self._gate(self.g2) * x
