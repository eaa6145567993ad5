from crdr_amd.utils.options import *  # noqa: F401,F403
from crdr_amd.utils import options as _m
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith('__')})
