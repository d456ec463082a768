"""Adaptive Information Transfer Multi-task model (API mirror of torch_rechub/models/multi_task/aitm.py:15-83).

Every task has a bottom MLP and a tower; task i > 0 replaces its bottom output by an attention over
{its own bottom output, info_gate(previous task's transferred state)} before the tower (all tasks are binary)."""
import math

import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer


class AITM(nn.Module):

    def __init__(self, features, n_task, bottom_params, tower_params_list):
        super().__init__()
        self.features, self.n_task = features, n_task
        self.input_dims = sum(f.embed_dim for f in features)
        self.embedding = EmbeddingLayer(features)
        hidden = bottom_params["dims"][-1]
        self.bottoms = nn.ModuleList(MLP(self.input_dims, output_layer=False, **bottom_params) for _ in range(n_task))
        self.towers = nn.ModuleList(MLP(hidden, **tower_params_list[i]) for i in range(n_task))
        self.info_gates = nn.ModuleList(MLP(hidden, output_layer=False, dims=[hidden]) for _ in range(n_task - 1))
        self.aits = nn.ModuleList(AttentionLayer(hidden) for _ in range(n_task - 1))

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        states = [bottom(embed_x) for bottom in self.bottoms]
        for i in range(1, self.n_task):
            info = self.info_gates[i - 1](states[i - 1])
            states[i] = self.aits[i - 1](torch.stack([states[i], info], dim=1))
        return torch.cat([torch.sigmoid(tower(h)) for h, tower in zip(states, self.towers)], dim=1)


class AttentionLayer(nn.Module):
    """Scaled dot-product self-weights over the 2 candidates: (B, 2, dim) -> (B, dim)."""

    def __init__(self, dim=32):
        super().__init__()
        self.dim = dim
        self.q_layer = nn.Linear(dim, dim, bias=False)
        self.k_layer = nn.Linear(dim, dim, bias=False)
        self.v_layer = nn.Linear(dim, dim, bias=False)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x):
        B, n, d = x.shape
        qkv = x.reshape(B * n, d) @ torch.cat([self.q_layer.weight, self.k_layer.weight, self.v_layer.weight]).t()
        q, k, v = qkv.view(B, n, 3, d).unbind(dim=2)
        a = self.softmax((q * k).sum(-1) / math.sqrt(self.dim))
        return (a.unsqueeze(-1) * v).sum(dim=1)
