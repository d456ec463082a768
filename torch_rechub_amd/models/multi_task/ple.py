"""Progressive Layered Extraction (API mirror of torch_rechub/models/multi_task/ple.py:13-116).

``n_level`` CGC layers; each keeps ``n_expert_specific`` experts per task plus ``n_expert_shared`` shared ones.  Task
t's gate mixes its own experts with the shared ones; below the last level a shared gate mixes all experts into the
input of the next level's shared experts.  Module names (``cgc_layers.i.{experts_specific, experts_shared,
gates_specific, gate_shared}``, ``towers``, ``predict_layers``) are the reference's."""
import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer
from ._common import gate_mix, run_heads, softmax_gate, task_heads


class PLE(nn.Module):

    def __init__(self, features, task_types, n_level, n_expert_specific, n_expert_shared, expert_params,
                 tower_params_list):
        super().__init__()
        self.features, self.task_types = features, task_types
        self.n_task, self.n_level = len(task_types), n_level
        self.input_dims = sum(f.embed_dim for f in features)
        self.embedding = EmbeddingLayer(features)
        self.cgc_layers = nn.ModuleList(
            CGC(level + 1, n_level, self.n_task, n_expert_specific, n_expert_shared, self.input_dims, expert_params)
            for level in range(n_level))
        self.towers, self.predict_layers = task_heads(expert_params["dims"][-1], task_types, tower_params_list)

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        state = [embed_x] * (self.n_task + 1)  # per-task streams + the shared stream (last)
        for layer in self.cgc_layers:
            state = layer(state)
        return run_heads(state[:self.n_task], self.towers, self.predict_layers)


class CGC(nn.Module):
    """Customized Gate Control layer: inputs / outputs are lists [task 0, ..., task n-1, shared]."""

    def __init__(self, cur_level, n_level, n_task, n_expert_specific, n_expert_shared, input_dims, expert_params):
        super().__init__()
        self.cur_level, self.n_level, self.n_task = cur_level, n_level, n_task
        self.n_expert_specific, self.n_expert_shared = n_expert_specific, n_expert_shared
        self.n_expert_all = n_expert_specific * n_task + n_expert_shared
        width = input_dims if cur_level == 1 else expert_params["dims"][-1]
        self.experts_specific = nn.ModuleList(
            MLP(width, output_layer=False, **expert_params) for _ in range(n_task * n_expert_specific))
        self.experts_shared = nn.ModuleList(MLP(width, output_layer=False, **expert_params) for _ in range(n_expert_shared))
        self.gates_specific = nn.ModuleList(softmax_gate(width, n_expert_specific + n_expert_shared) for _ in range(n_task))
        if cur_level < n_level:
            self.gate_shared = softmax_gate(width, self.n_expert_all)

    def forward(self, x_list):
        k = self.n_expert_specific
        own = [self.experts_specific[t * k + e](x_list[t]) for t in range(self.n_task) for e in range(k)]
        shared = [expert(x_list[-1]) for expert in self.experts_shared]
        outs = []
        for t, gate in enumerate(self.gates_specific):
            outs.append(gate_mix(gate(x_list[t]), torch.stack(own[t * k:(t + 1) * k] + shared, dim=1)))
        if self.cur_level < self.n_level:
            outs.append(gate_mix(self.gate_shared(x_list[-1]), torch.stack(own + shared, dim=1)))
        return outs
