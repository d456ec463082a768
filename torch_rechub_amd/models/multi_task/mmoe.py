"""Multi-gate Mixture-of-Experts (API mirror of torch_rechub/models/multi_task/mmoe.py:13-58).

One fused gather feeds ``n_expert`` expert MLPs and one softmax gate per task; task t's tower reads
sum_e gate_t[e] * expert_e (one batched product per task instead of a broadcast multiply + reduction)."""
import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer
from ._common import gate_mix, run_heads, softmax_gate, task_heads


class MMOE(nn.Module):

    def __init__(self, features, task_types, n_expert, expert_params, tower_params_list):
        super().__init__()
        self.features, self.task_types = features, task_types
        self.n_task, self.n_expert = len(task_types), n_expert
        self.embedding = EmbeddingLayer(features)
        self.input_dims = sum(f.embed_dim for f in features)
        self.experts = nn.ModuleList(MLP(self.input_dims, output_layer=False, **expert_params) for _ in range(n_expert))
        self.gates = nn.ModuleList(softmax_gate(self.input_dims, n_expert) for _ in range(self.n_task))
        self.towers, self.predict_layers = task_heads(expert_params["dims"][-1], task_types, tower_params_list)

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        experts = torch.stack([expert(embed_x) for expert in self.experts], dim=1)  # (B, E, H)
        mixed = [gate_mix(gate(embed_x), experts) for gate in self.gates]
        return run_heads(mixed, self.towers, self.predict_layers)
