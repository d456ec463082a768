"""Activations (API mirror of torch_rechub/basic/activation.py:5-54).

``Dice`` keeps the reference's exact (non-paper) definition, SURVEY Q5: statistics are taken ACROSS
NEURONS per row (dim=1) and the "variance" is the SUM of (x-mean)^2 + eps over the row.
"""
import torch
from torch import nn

from .. import ops


class Dice(nn.Module):
    """p = sigmoid((x - mean_row) / sqrt(sum_row((x-mean_row)^2 + eps)));  out = p*x + (1-p)*alpha*x."""

    def __init__(self, epsilon=1e-3):
        super().__init__()
        self.epsilon = epsilon
        self.alpha = nn.Parameter(torch.randn(1))

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 2048:
            return ops.dice(x, self.alpha, self.epsilon)  # one pass instead of ten elementwise kernels
        ops.require_hip(x)
        mu = x.mean(dim=1, keepdim=True)
        c = x - mu
        var = (c * c + self.epsilon).sum(dim=1, keepdim=True)
        p = torch.sigmoid(c / torch.sqrt(var))
        return p * x + (1 - p) * self.alpha * x


def activation_layer(act_name):
    """Build an activation module from its name or class (reference activation.py:28-54)."""
    if isinstance(act_name, str):
        key = act_name.lower()
        table = {
            "sigmoid": nn.Sigmoid,
            "relu": lambda: nn.ReLU(inplace=True),
            "dice": Dice,
            "prelu": nn.PReLU,
            "softmax": lambda: nn.Softmax(dim=1),
            "leakyrelu": nn.LeakyReLU,
        }
        if key not in table:
            raise NotImplementedError(f"activation {act_name!r} is not supported")
        return table[key]()
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError
