from crdr_amd.utils.path import *  # noqa: F401,F403
from crdr_amd.utils import path as _m
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith('__')})
