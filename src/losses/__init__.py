from crdr_amd.losses import build_loss  # noqa: F401
