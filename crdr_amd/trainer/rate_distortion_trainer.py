"""Stage 1 / 2 trainer: rate + distortion (+ LPIPS), Adam on the generator, a second Adam on the quantiles
(src/trainer/rate_distortion_trainer.py:17-163)."""
from __future__ import annotations

import os
from copy import deepcopy
from typing import Dict, Optional, Tuple

import torch

from crdr_amd.losses import build_loss
from crdr_amd.utils.path import PathHandler
from crdr_amd.utils.registry import TRAINER_REGISTRY

from . import dist as D
from .base_trainer import BaseTrainer
from .optimizer import build_optimizer, build_scheduler


@TRAINER_REGISTRY.register()
class RateDistortionTrainer(BaseTrainer):
    def _set_losses(self):
        lo = deepcopy(self.opt.loss)
        self.distortion_loss = build_loss(lo.distortion_loss, loss_name="distortion_loss")
        self.rate_loss = build_loss(lo.rate_loss, loss_name="rate_loss")
        self.perceptual_loss = build_loss(lo.perceptual_loss, loss_name="perceptual_loss").to(self.device) if lo.get("perceptual_loss") else None

    def _set_optimizer_scheduler(self):
        params, aux = self.comp_model.separate_aux_parameters()
        oo = deepcopy(self.opt.optim)
        self.g_optimizer = build_optimizer(params, oo.g_optimizer)
        self.g_scheduler = build_scheduler(self.g_optimizer, oo.g_scheduler) if oo.get("g_scheduler") else None
        self.aux_optimizer = build_optimizer(aux, oo.aux_optimizer) if len(aux) > 0 else None
        self.clip_max_norm = float(oo.get("clip_max_norm", 0) or 0)

    def run_comp_model(self, data_dict: Dict) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict]:
        out = self.comp_model.run_model(**data_dict)
        return out.pop("real_images"), out.pop("fake_images"), out.pop("bpp"), out

    def _rate_kwargs(self, other: Dict) -> Dict:
        """Under data parallelism the lambda_A / lambda_B switch must see the global mean of qbpp."""
        if D.is_dist() and "qbpp" in other:
            return {"qbpp_mean": D.all_reduce_scalars_mean(other["qbpp"].detach().mean())}
        return {}

    def _step_generator(self, l_total) -> None:
        """backward -> (all-reduce) -> global-norm clip folded into the fused Adam -> scheduler."""
        l_total.backward()
        D.all_reduce_mean_(self.g_optimizer.flat_grads())
        sq = None
        if self.clip_max_norm:
            sq = self.g_optimizer.grad_sqnorm()
            if self.aux_optimizer is not None:  # clip_grad_norm_ runs over comp_model.parameters(): quantiles included
                sq = sq + self.aux_optimizer.grad_sqnorm()
        self.g_optimizer.step(sqnorm=sq, max_norm=self.clip_max_norm)

    def optimize_parameters(self, current_iter: int, data_dict: Dict):
        log: Dict = {}
        self.g_optimizer.zero_grad()
        if self.aux_optimizer:
            self.aux_optimizer.zero_grad()
        real, fake, bpp, other = self.run_comp_model(data_dict)
        log["qbpp"] = other.get("qbpp", -1)
        g = {"distortion": self.distortion_loss(real, fake, **other),
             "rate": self.rate_loss(bpp, **other, **self._rate_kwargs(other), current_iter=current_iter)}
        if self.perceptual_loss:
            g["perceptual"] = self.perceptual_loss(real, fake)
        l_total = sum(g.values())
        bad = self.check_loss_nan_inf(l_total)
        if D.any_rank_true(bool(bad), l_total.device):
            self.logger.warning(f"iter{current_iter}: skipped because loss is {bad or 'bad on another rank'}")
            return None
        self._step_generator(l_total)
        log.update(g)
        if self.g_scheduler:
            self.g_scheduler.step()
        if self.aux_optimizer:
            log["aux"] = self.optimize_aux_parameters()
        return log

    def optimize_aux_parameters(self):
        self.aux_optimizer.zero_grad()
        aux = self.comp_model.aux_loss()
        aux.backward()
        D.all_reduce_mean_(self.aux_optimizer.flat_grads())
        self.aux_optimizer.step()
        return aux

    def _training_state(self) -> Dict:
        st = {"g_optimizer": self.g_optimizer}
        if self.aux_optimizer:
            st["aux_optimizer"] = self.aux_optimizer
        if self.g_scheduler:
            st["g_scheduler"] = self.g_scheduler
        return st

    def save(self, current_iter: int):
        self.model_saver.save({"comp_model": self.comp_model}, "comp_model", current_iter, keep=True)
        self.model_saver.save(self._training_state(), "training_state", current_iter, keep=self.opt.get("keep_training_state", False))

    def _load_checkpoint(self, exp: str, itr: int, load_optimizer: bool = True, load_scheduler: bool = True,
                         new_g_lr: Optional[float] = None, strict: bool = True, **kwargs) -> None:
        ph = PathHandler(self.opt.path.ckpt_root, exp)
        mp = ph.get_ckpt_path("comp_model", itr)
        assert os.path.exists(mp), mp
        self.comp_model.load_state_dict(torch.load(mp, map_location="cpu")["comp_model"], strict=strict)
        if not load_optimizer:
            return
        op = ph.get_ckpt_path("training_state", itr)
        assert os.path.exists(op), op
        st = torch.load(op, map_location="cpu")
        self.g_optimizer.load_state_dict(st["g_optimizer"])
        if new_g_lr is not None:
            self.update_learning_rate(self.g_optimizer, new_g_lr)
        if self.g_scheduler and load_scheduler:
            self.g_scheduler.load_state_dict(st["g_scheduler"])
        if self.aux_optimizer:
            self.aux_optimizer.load_state_dict(st["aux_optimizer"])
