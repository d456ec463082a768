"""Adaptive Information Transfer Multi-task model (API mirror of torch_rechub/models/multi_task/aitm.py:15-83).

Every task has a bottom MLP and a tower; task i > 0 replaces its bottom output by an attention over
{its own bottom output, info_gate(previous task's transferred state)} before the tower (all tasks are binary).
Module names (``bottoms``, ``towers``, ``info_gates``, ``aits`` with ``q_layer / k_layer / v_layer``) are the
reference's; the three attention projections run as one product."""
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
        tasks, transfers = range(n_task), range(n_task - 1)
        self.bottoms = nn.ModuleList(MLP(self.input_dims, output_layer=False, **bottom_params) for _ in tasks)
        self.towers = nn.ModuleList(MLP(hidden, **tower_params_list[t]) for t in tasks)
        self.info_gates = nn.ModuleList(MLP(hidden, output_layer=False, dims=[hidden]) for _ in transfers)
        self.aits = nn.ModuleList(AttentionLayer(hidden) for _ in transfers)

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        state, probs = None, []
        for t in range(self.n_task):
            own = self.bottoms[t](embed_x)
            if t > 0:  # what the previous task knows, gated, competes with the task's own view
                own = self.aits[t - 1](torch.stack([own, self.info_gates[t - 1](state)], dim=1))
            state = own
            probs.append(torch.sigmoid(self.towers[t](own)))
        return torch.cat(probs, dim=1)


class AttentionLayer(nn.Module):
    """Scaled dot-product self-weights over the 2 candidates: (B, 2, dim) -> (B, dim)."""

    def __init__(self, dim=32):
        super().__init__()
        self.dim = dim
        self.q_layer, self.k_layer, self.v_layer = (nn.Linear(dim, dim, bias=False) for _ in range(3))
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x):
        B, n, d = x.shape
        qkv = x.reshape(B * n, d) @ torch.cat([self.q_layer.weight, self.k_layer.weight, self.v_layer.weight]).t()
        q, k, v = qkv.view(B, n, 3, d).unbind(dim=2)
        a = self.softmax((q * k).sum(-1) / math.sqrt(self.dim))
        return (a.unsqueeze(-1) * v).sum(dim=1)
