"""Stage-3 CRDR training step (src/trainer/multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:13-114).

Per iteration (q = rate index, beta = realism weight, both sampled once per batch):
  G: x_hat = G(x; q, beta);  x_hr = G(x; q+1, beta) without grad (or x itself at the top rate);
     L = 150 MSE + lambda(q) bpp + beta * (w_lpips LPIPS(x, x_hat) + adv),
     adv = w_gan/2 [BCE(D(x_hr) - D(x_hat), 0) + BCE(D(x_hat) - D(x_hr), 1)]   (D frozen);
     clip 1.0, Adam; aux Adam on the quantiles.
  D: L_D = 1/2 BCE(D(x) - D(x_hat)^, 1) + 1/2 BCE(D(x_hat)^ - D(x)^, 0), ^ = detached; Adam.
The reference evaluates D(x_hat) three times per iteration with unchanged discriminator weights; here the two
discriminator-phase evaluations share one forward (identical values, gradients sum to the same total).
Data parallel: every rank draws the same (q, beta) from a shared seeded generator so exactly one
sub-discriminator is active per iteration and gradients average like one big batch."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from crdr_amd.utils.registry import TRAINER_REGISTRY

from . import dist as D
from .multirate_hr_rgan_rate_distortion_trainer import MultirateHighRateRGANRateDistortionTrainer


@TRAINER_REGISTRY.register()
class MultirateBetaCondHrrGanRateDistortionTrainer(MultirateHighRateRGANRateDistortionTrainer):
    def __init__(self, opt, relative_score_rate_delta=1) -> None:
        super().__init__(opt, relative_score_rate_delta)
        seed = int(opt.get("cond_seed", 0))
        self._q_gen = torch.Generator().manual_seed(seed)
        self._b_rng = np.random.default_rng(seed)

    def _sample_conditions(self, data_dict: Dict) -> None:
        if "rate_ind" not in data_dict or data_dict["rate_ind"] is None:
            if D.is_dist():
                data_dict["rate_ind"] = torch.randint(self.rate_level, (1,), generator=self._q_gen)
            else:
                data_dict["rate_ind"] = self.comp_model.sample_rate_ind()
        if "beta" not in data_dict or data_dict["beta"] is None:
            if D.is_dist():
                data_dict["beta"] = self.comp_model.max_beta * (float(self._b_rng.integers(0, 101)) / 100.0)
            else:
                data_dict["beta"] = self.comp_model.sample_beta()

    def optimize_parameters(self, current_iter: int, data_dict: Dict) -> Dict:
        log: Dict = {}
        data_dict = dict(data_dict)
        self._sample_conditions(data_dict)
        # ------------------------------------------------------------------ G
        self.discriminator.requires_grad_(False)
        self.g_optimizer.zero_grad()
        if self.aux_optimizer:
            self.aux_optimizer.zero_grad()
        real, fake, bpp, other = self.run_comp_model(data_dict)
        rate_ind, beta = other["rate_ind"], other["beta"]
        log["qbpp"] = other.get("qbpp", -1)

        high = rate_ind + self.relative_score_rate_delta
        if float(high) > self.rate_level - 1:
            relative = real
        else:
            hr = dict(data_dict)
            hr["rate_ind"], hr["beta"] = high, beta
            with torch.no_grad():
                _, relative, _, _ = self.run_comp_model(hr)

        dist_loss = self.distortion_loss(real, fake, **other)
        rate_loss = self.rate_loss(bpp, **other, **self._rate_kwargs(other), current_iter=current_iter)
        assert self.perceptual_loss
        percep = self.perceptual_loss(real, fake)
        with torch.no_grad():
            real_d = self.discriminator(relative.detach(), **other)
        fake_g = self.discriminator(fake, **other)
        adv = (self.gan_loss.forward_diff(real_d, fake_g, is_real=False, is_disc=False)
               + self.gan_loss.forward_diff(fake_g, real_d, is_real=True, is_disc=False)) / 2
        g = {"distortion": dist_loss, "rate": rate_loss, "perceptual": percep, "adv": adv}
        l_total = dist_loss + rate_loss + beta * (percep + adv)

        bad = self.check_loss_nan_inf(l_total)
        if D.any_rank_true(bool(bad), l_total.device):
            self.logger.warning(f"iter{current_iter}: skipped because loss is {bad or 'bad on another rank'}")
            return None
        self._step_generator(l_total)
        if self.aux_optimizer:
            log["aux"] = self.optimize_aux_parameters()
        if self.g_scheduler:
            self.g_scheduler.step()
        log.update(g)
        # ------------------------------------------------------------------ D
        self.discriminator.requires_grad_(True)
        self.d_optimizer.zero_grad()
        fake_d = self.discriminator(fake.detach(), **other)
        real_d = self.discriminator(real, **other)
        l_d_real = self.gan_loss.forward_diff(real_d, fake_d.detach(), is_real=True, is_disc=True) * 0.5
        l_d_fake = self.gan_loss.forward_diff(fake_d, real_d.detach(), is_real=False, is_disc=True) * 0.5
        self._step_discriminator(l_d_real + l_d_fake)
        log.update({"d_real": l_d_real, "d_fake": l_d_fake, "d_total": l_d_real + l_d_fake,
                    "out_d_real": real_d.detach().mean(), "out_d_fake": fake_d.detach().mean()})
        return log
