"""Pieces the multi-task models share: the per-task heads and the gated mixture of expert outputs."""
import torch
from torch import nn

from ...basic.layers import MLP, PredictionLayer


def task_heads(in_dim, task_types, tower_params_list):
    """(towers, predict_layers): one MLP (+ output Linear) and one sigmoid / identity per task."""
    towers = nn.ModuleList(MLP(in_dim, **tower_params_list[i]) for i in range(len(task_types)))
    heads = nn.ModuleList(PredictionLayer(t) for t in task_types)
    return towers, heads


def run_heads(inputs, towers, heads):
    """(B, n_task): task i's tower and prediction layer over ``inputs[i]`` (one shared tensor when not a list)."""
    if not isinstance(inputs, (list, tuple)):
        inputs = [inputs] * len(towers)
    return torch.cat([head(tower(h)) for h, tower, head in zip(inputs, towers, heads)], dim=1)


def gate_mix(gate, experts):
    """sum_e gate[:, e] * experts[:, e, :]   (gate (B, E) softmax weights, experts (B, E, H)) -> (B, H)."""
    return torch.bmm(gate.unsqueeze(1), experts).squeeze(1)


def softmax_gate(in_dim, n_out):
    """Linear -> BatchNorm1d -> Softmax(dim=1) -> Dropout(0): the reference spells a gate as an MLP without an output
    layer and activation "softmax" (mmoe.py:33, ple.py:82-84), so the checkpoint keys are an MLP's."""
    return MLP(in_dim, output_layer=False, dims=[n_out], activation="softmax")
