import torch.nn as nn


class BaseDiscriminator(nn.Module):
    def __init__(self) -> None:
        super().__init__()
