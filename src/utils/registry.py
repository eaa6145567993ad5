from crdr_amd.utils.registry import *  # noqa: F401,F403
from crdr_amd.utils import registry as _m
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith('__')})
