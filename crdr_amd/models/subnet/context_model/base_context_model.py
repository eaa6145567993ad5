import torch.nn as nn


class BaseContextModel(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
