"""`freerec.models.nn`: small modules the scripts use (DeepFM/main.py:52,147 `Unsqueeze(1)`)."""
import torch.nn as nn


class Unsqueeze(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.unsqueeze(self.dim)


class Squeeze(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.squeeze(self.dim)
