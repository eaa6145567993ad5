from crdr_amd.trainer import build_trainer  # noqa: F401
