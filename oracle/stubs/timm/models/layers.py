import torch.nn as nn
from torch.nn.init import trunc_normal_  # noqa: F401


class DropPath(nn.Module):
    """Stochastic depth: identity in eval mode (the only mode the oracle uses)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training, 'stub DropPath supports eval mode only'
        return x
