"""CRDR (stage 3 / inference): rate index q and realism weight beta as run-time knobs
(src/models/comp_model/beta_cond_interpca_hyperprior_model.py:18-208 and
beta_cond_interpca_hyperprior_charm_model.py:18-149)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import numpy as np
from torch import Tensor

from crdr_amd.utils.registry import MODEL_REGISTRY

from .interpca_hyperprior_charm_model import InterpCaHyperpriorCharmModel


@MODEL_REGISTRY.register()
class BetaCondInterpCaHyperpriorCharmModel(InterpCaHyperpriorCharmModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.max_beta: float = opt.subnet.decoder.max_beta

    def sample_beta(self) -> float:
        return self.max_beta * (float(np.random.randint(0, 101)) / 100.0)

    def _decode(self, y_hat, rate_ind=None, beta=0.0, **cond):
        return self.decoder(y_hat, rate_ind, beta=beta)

    def _extra_outputs(self, rate_ind=None, beta=None, **cond) -> Dict:
        return {"rate_ind": rate_ind, "beta": beta}

    def run_model(self, real_images: Tensor, rate_ind: Optional[Union[float, Tensor]] = None, beta: Optional[float] = None,
                  is_train: bool = True, noise=None) -> dict:
        if beta is None:
            if not is_train:
                raise ValueError('"beta" must be specified if is_train=False')
            beta = self.sample_beta()
        return super().run_model(real_images, rate_ind=rate_ind, is_train=is_train, noise=noise, beta=beta)

    def decompress(self, string_list: List, beta: float = 0.0):
        return super().decompress(string_list, beta=beta)

    def _validation_conditions(self, beta: Optional[float] = None, **kw) -> List[Tuple[str, Dict]]:
        beta = self.max_beta / 2.0 if beta is None else beta
        return [(f"_{q + 1}", {"rate_ind": float(q), "beta": beta}) for q in range(self.rate_level)]
