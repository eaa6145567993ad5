import torch.nn as nn


class BaseEncoder(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.num_downscale = None  # set by subclasses
        self.latent_ch = None


class BaseDecoder(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
