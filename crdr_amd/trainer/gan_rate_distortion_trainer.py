"""Adds the discriminator, its optimiser and the GAN loss to the rate-distortion trainer
(src/trainer/gan_rate_distortion_trainer.py:17-227). The plain (non-relativistic) objective of this class is an
ablation in the reference; the CRDR recipe uses the multirate relativistic subclass."""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import torch

from crdr_amd.losses import build_loss
from crdr_amd.models.discriminator import build_discriminator
from crdr_amd.utils.path import PathHandler
from crdr_amd.utils.registry import TRAINER_REGISTRY

from . import dist as D
from .optimizer import build_optimizer, build_scheduler
from .rate_distortion_trainer import RateDistortionTrainer


@TRAINER_REGISTRY.register()
class GANRateDistortionTrainer(RateDistortionTrainer):
    def _set_models(self) -> None:
        super()._set_models()
        self.discriminator = build_discriminator(self.opt.discriminator).to(self.device)
        D.broadcast_module_(self.discriminator)
        self.discriminator.train()

    def _set_losses(self) -> None:
        super()._set_losses()
        self.gan_loss = build_loss(deepcopy(self.opt.loss).gan_loss, loss_name="gan_loss")

    def _set_optimizer_scheduler(self) -> None:
        super()._set_optimizer_scheduler()
        oo = deepcopy(self.opt.optim)
        self.d_optimizer = build_optimizer(dict(self.discriminator.named_parameters()), oo.d_optimizer)
        if hasattr(self.discriminator, "subD_list") and len(self.d_optimizer.param_groups) == 1:
            # one partition per sub-discriminator: only the one a step ran is zeroed / all-reduced / updated, like
            # torch.optim.Adam skipping the parameters whose .grad is None (module_list_discriminator.py:26-30)
            self.d_optimizer.set_partitions(self.discriminator.subD_list)
        self.d_scheduler = build_scheduler(self.d_optimizer, oo.d_scheduler) if oo.get("d_scheduler") else None

    def _finish_step(self, current_iter: int, log):
        vals = super()._finish_step(current_iter, log)
        if vals is not None and self.d_scheduler:
            self.d_scheduler.step()
        return vals

    def _training_state(self) -> Dict:
        st = super()._training_state()
        st["d_optimizer"] = self.d_optimizer
        if self.d_scheduler:
            st["d_scheduler"] = self.d_scheduler
        return st

    def save(self, current_iter: int) -> None:
        self.model_saver.save({"comp_model": self.comp_model}, "comp_model", current_iter, keep=True)
        self.model_saver.save({"discriminator": self.discriminator}, "discriminator", current_iter, keep=self.opt.get("keep_discriminator", False))
        self.model_saver.save(self._training_state(), "training_state", current_iter, keep=self.opt.get("keep_training_state", False))

    def _load_checkpoint(self, exp: str, itr: int, load_optimizer: bool = True, load_discriminator: bool = True,
                         load_scheduler: bool = True, new_g_lr: Optional[float] = None, new_d_lr: Optional[float] = None,
                         strict: bool = True, **kwargs) -> None:
        super()._load_checkpoint(exp, itr, load_optimizer=load_optimizer, load_scheduler=load_scheduler, new_g_lr=new_g_lr, strict=strict)
        if not load_discriminator:
            return
        ph = PathHandler(self.opt.path.ckpt_root, exp)
        self.discriminator.load_state_dict(torch.load(ph.get_ckpt_path("discriminator", itr), map_location="cpu")["discriminator"], strict=strict)
        if load_optimizer:
            st = torch.load(ph.get_ckpt_path("training_state", itr), map_location="cpu")
            self.d_optimizer.load_state_dict(st["d_optimizer"])
            if new_d_lr is not None:
                self.update_learning_rate(self.d_optimizer, new_d_lr)
            if self.d_scheduler and load_scheduler:
                self.d_scheduler.load_state_dict(st["d_scheduler"])
