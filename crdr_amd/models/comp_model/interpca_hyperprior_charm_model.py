"""Stage-2 model: + interpolation channel attention for variable rate (rate index sampled per batch)
(src/models/comp_model/interpca_hyperprior_model.py:19-224; interpca_hyperprior_charm_model.py)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor

from crdr_amd.utils.codec_utils import MultiRateHeaderHandler
from crdr_amd.utils.registry import MODEL_REGISTRY

from .hyperprior_charm_model import HyperpriorCharmModel


@MODEL_REGISTRY.register()
class InterpCaHyperpriorCharmModel(HyperpriorCharmModel):
    def __init__(self, opt):
        self.rate_level = opt.subnet.encoder.rate_level
        assert opt.subnet.encoder.rate_level == opt.subnet.decoder.rate_level
        super().__init__(opt)
        if opt.get("batch_rate_ind_sample", False):
            raise NotImplementedError("batch_rate_ind_sample is not supported yet.")
        self.batch_rate_ind_sample = False

    def sample_rate_ind(self, num_sample: int = 1) -> Tensor:
        return torch.randint(self.rate_level, (num_sample,))

    def _encode(self, x, rate_ind=None, **cond):
        return self.encoder(x, rate_ind)

    def _decode(self, y_hat, rate_ind=None, **cond):
        return self.decoder(y_hat, rate_ind)

    def _extra_outputs(self, rate_ind=None, **cond) -> Dict:
        return {"rate_ind": rate_ind}

    def run_model(self, real_images, rate_ind: Optional[Union[Tensor, float]] = None, is_train: bool = True, noise=None, **cond):
        if rate_ind is None:
            if not is_train:
                raise ValueError('"rate_ind" must be specified if is_train=False')
            rate_ind = self.sample_rate_ind()
        return super().run_model(real_images, is_train=is_train, noise=noise, rate_ind=rate_ind, **cond)

    def _make_header_handler(self):
        return MultiRateHeaderHandler(use_non_zero_ind=False)

    def _header_encode(self, size, y_hat, rate_ind=None, **cond) -> bytes:
        return self.header_handler.encode(size, y_hat, rate_ind=rate_ind)

    def _header_cond(self, header: Dict) -> Dict:
        return {"rate_ind": header["rate_ind"]}

    def compress(self, real_images: Tensor, rate_ind: Union[Tensor, float]) -> Dict:
        return super().compress(real_images, rate_ind=rate_ind)

    def _validation_conditions(self, **kw) -> List[Tuple[str, Dict]]:
        return [(f"_{q + 1}", {"rate_ind": float(q)}) for q in range(self.rate_level)]
