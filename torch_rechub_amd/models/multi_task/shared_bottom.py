"""Shared-Bottom multi-task model (API mirror of torch_rechub/models/multi_task/shared_bottom.py:13-45).

One flattened fused gather -> one bottom MLP (no output layer) -> a tower + sigmoid / identity per task -> (B, n_task).
Attribute names are the reference's (``embedding``, ``bottom_mlp``, ``towers``, ``predict_layers``)."""
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer
from ._common import run_heads, task_heads


class SharedBottom(nn.Module):

    def __init__(self, features, task_types, bottom_params, tower_params_list):
        super().__init__()
        self.features, self.task_types = features, task_types
        self.embedding = EmbeddingLayer(features)
        self.bottom_dims = sum(f.embed_dim for f in features)
        self.bottom_mlp = MLP(self.bottom_dims, **{**bottom_params, "output_layer": False})
        self.towers, self.predict_layers = task_heads(bottom_params["dims"][-1], task_types, tower_params_list)

    def forward(self, x):
        shared = self.bottom_mlp(self.embedding(x, self.features, squeeze_dim=True))
        return run_heads(shared, self.towers, self.predict_layers)
