from crdr_amd.utils.model_saver import *  # noqa: F401,F403
from crdr_amd.utils import model_saver as _m
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith('__')})
