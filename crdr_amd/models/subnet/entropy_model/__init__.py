from .entropy_bottleneck import EntropyBottleneck, SteEntropyBottleneck  # noqa: F401
from .gaussian_conditional import GaussianMeanScaleConditional  # noqa: F401
from .ste_gaussian_conditional import SteGaussianMeanScaleConditional  # noqa: F401
