from .build_optimizer_scheduler import build_optimizer, build_scheduler  # noqa: F401
