from crdr_amd.models import build_comp_model, build_trained_comp_model  # noqa: F401
from crdr_amd.models.subnet import build_subnet  # noqa: F401
from crdr_amd.models.discriminator import build_discriminator  # noqa: F401
