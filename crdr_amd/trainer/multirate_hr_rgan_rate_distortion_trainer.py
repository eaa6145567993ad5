"""Multi-rate "high-rate relativistic GAN" trainer base: the relative score image is the no-grad reconstruction
one rate level up (src/trainer/multirate_hr_rgan_rate_distortion_trainer.py:11-14)."""
from __future__ import annotations

from crdr_amd.utils.registry import TRAINER_REGISTRY

from .gan_rate_distortion_trainer import GANRateDistortionTrainer


@TRAINER_REGISTRY.register()
class MultirateHighRateRGANRateDistortionTrainer(GANRateDistortionTrainer):
    def __init__(self, opt, relative_score_rate_delta=1) -> None:
        super().__init__(opt)
        self.rate_level = self.comp_model.rate_level
        self.relative_score_rate_delta = relative_score_rate_delta
