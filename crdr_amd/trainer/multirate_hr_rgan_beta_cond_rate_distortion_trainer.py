"""Stage-3 CRDR training step (src/trainer/multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:13-114).

Per iteration (q = rate index, beta = realism weight, both sampled once per batch):
  G: x_hat = G(x; q, beta);  x_hr = G(x; q+1, beta) without grad (or x itself at the top rate);
     L = 150 MSE + lambda(q) bpp + beta * (w_lpips LPIPS(x, x_hat) + adv),
     adv = w_gan/2 [BCE(D(x_hr) - D(x_hat), 0) + BCE(D(x_hat) - D(x_hr), 1)]   (D frozen);
     clip 1.0, Adam; aux Adam on the quantiles.
  D: L_D = 1/2 BCE(D(x) - D(x_hat)^, 1) + 1/2 BCE(D(x_hat)^ - D(x)^, 0), ^ = detached; Adam.
The reference evaluates D(x_hat) three times per iteration with unchanged discriminator weights; here the two
discriminator-phase evaluations share one forward (identical values, gradients sum to the same total).
One documented deviation: with a spectral-norm discriminator (HiFiCDiscriminator, not used by any shipped config) every
training-mode forward runs one power iteration, so the reference's 5 forwards per step advance u / v five times where
this step advances them three times (hr, fake, [fake; real]) -- same fixed point, slightly slower convergence to it.
Data parallel: every rank draws the same (q, beta) from a shared seeded generator so exactly one
sub-discriminator is active per iteration and gradients average like one big batch."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from crdr_amd.hip import chain as _chain
from crdr_amd.hip import ops as _ops
from crdr_amd.utils.registry import TRAINER_REGISTRY

from . import dist as D
from .multirate_hr_rgan_rate_distortion_trainer import MultirateHighRateRGANRateDistortionTrainer


@TRAINER_REGISTRY.register()
class MultirateBetaCondHrrGanRateDistortionTrainer(MultirateHighRateRGANRateDistortionTrainer):
    def __init__(self, opt, relative_score_rate_delta=1) -> None:
        super().__init__(opt, relative_score_rate_delta)
        # `reuse_d_forward` (default on): the discriminator phase takes D(x_hat) from the generator phase's forward instead of evaluating it again
        self.reuse_d_forward = bool(opt.get("reuse_d_forward", __import__("os").environ.get("CRDR_REUSE_D_FORWARD", "1") != "0"))
        seed = int(opt.get("cond_seed", 0))
        self._q_gen = torch.Generator().manual_seed(seed)
        self._b_rng = np.random.default_rng(seed)

    def _conditions(self, data_dict: Dict):
        """Draw (q, beta) for this iteration (reference: one draw per batch, interpca_hyperprior_model.py:43-45 and
        beta_cond_interpca_hyperprior_model.py:41-46).  q selects the graph; beta enters through device buffers."""
        q = data_dict.get("rate_ind")
        if q is None:
            q = torch.randint(self.rate_level, (1,), generator=self._q_gen) if D.is_dist() else self.comp_model.sample_rate_ind()
        q = int(q.item()) if isinstance(q, torch.Tensor) else int(q)
        beta = data_dict.get("beta")
        if beta is None:
            beta = (self.comp_model.max_beta * (float(self._b_rng.integers(0, 101)) / 100.0)) if D.is_dist() else self.comp_model.sample_beta()
        beta = float(beta)
        dec = self.comp_model.decoder
        if getattr(dec, "static_embed", None) is None:
            dec.static_embed = torch.zeros(1, dec.embed.embed(0.0).shape[1], 1, 1, device=self.device)
            self._beta_t = torch.zeros(1, device=self.device)
        dec.static_embed.copy_(dec.embed.embed(beta).reshape(dec.static_embed.shape))
        self._beta_t.fill_(beta)
        return {"rate_ind": q, "beta": beta}, ("s3", q)

    # ------------------------------------------------------------------ G phase: forward, losses, backward
    def _g_forward(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        # (no parameter changes inside: the generator and the discriminator run twice each on the same weights, and the second pass
        # reuses the Winograd-transformed filters of the first)
        with _ops.filter_scope():
            return self._g_forward_body(real, cond, noise, current_iter)

    def _g_forward_body(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        q, beta_t = cond["rate_ind"], self._beta_t
        self.discriminator.requires_grad_(False)
        self.g_optimizer.zero_grad()
        if self.aux_optimizer:
            self.aux_optimizer.zero_grad()
        data = {"real_images": real, "rate_ind": float(q), "beta": cond["beta"]}
        if noise is not None:
            data["noise"] = noise
        real_p, fake, bpp, other = self.run_comp_model(data)
        other["rate_ind"] = q
        if q + self.relative_score_rate_delta > self.rate_level - 1:
            relative = real_p
        else:
            # high-rate reconstruction at q + delta (reference :42-47 runs the full model under no_grad and keeps only
            # fake_images): the rate terms of that pass are dead code, reconstruct() does not evaluate them
            hr = {k: v for k, v in data.items() if k != "noise"}
            hr["rate_ind"] = float(q + self.relative_score_rate_delta)
            relative = self.comp_model.reconstruct(**hr)["fake_images"]
        dist_loss = self.distortion_loss(real_p, fake, **other)
        assert self.perceptual_loss
        percep = self.perceptual_loss(real_p, fake)
        # D(x_hat) is needed again by the discriminator phase with the same weights (the reference evaluates it three times per iteration,
        # ...beta_cond_...trainer.py:52, 92, 98): a chain-built discriminator keeps this pass's activations for that phase's backward.  At the top
        # rate the relativistic reference IS the real image, so D(x) of this phase serves the discriminator phase as well.
        keep = self.reuse_d_forward

        def d_keep(img, on: bool):
            if on:
                _chain.KEEP_HANDLES = []
            try:
                out = self.discriminator(img, **other)
            finally:
                handles, _chain.KEEP_HANDLES = _chain.KEEP_HANDLES, None
            return out, (handles[0] if (on and handles is not None and len(handles) == 1) else None)
        with torch.no_grad():
            real_d, d_real_handle = d_keep(relative.detach(), keep and relative is real_p)
        fake_g, d_handle = d_keep(fake, keep)
        adv = (self.gan_loss.forward_diff(real_d, fake_g, is_real=False, is_disc=False)
               + self.gan_loss.forward_diff(fake_g, real_d, is_real=True, is_disc=False)) / 2
        return {"bpp": bpp, "other": other, "terms": {"distortion": dist_loss, "perceptual": percep, "adv": adv},
                "nonrate": dist_loss + beta_t * (percep + adv), "extra": {"real": real_p, "fake": fake.detach(), "q": q, "d_handle": d_handle, "d_real_handle": d_real_handle}}

    def _seg_generator(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        f = self._g_forward(real, cond, noise, current_iter)
        other, t = f["other"], f["terms"]
        rate_loss = self.rate_loss(f["bpp"], **other, **self._rate_kwargs(other), current_iter=current_iter)
        l_total = t["distortion"] + rate_loss + self._beta_t * (t["perceptual"] + t["adv"])
        l_total.backward()
        self._flush_wgrads("g")
        return {"losses": {"distortion": t["distortion"], "rate": rate_loss, "perceptual": t["perceptual"], "adv": t["adv"]},
                "bad": self._bad_flag(l_total), "qbpp": other.get("qbpp", None), **f["extra"]}

    # ------------------------------------------------------------------ D forward + backward (needs only x and x̂)
    def _seg_dfwdbwd(self, ctx: Dict) -> Dict:
        """The reference steps G before it runs D on (x, x̂.detach()) (…beta_cond_…trainer.py:70-100); D's forward and
        backward do not read G's parameters, so running them BEFORE the G update is the same computation -- and lets
        them overlap the all-reduce of G's gradients."""
        q = ctx["q"]
        self.discriminator.requires_grad_(True)
        self.d_optimizer.zero_grad(partitions=self._d_parts(q))
        if ctx.get("d_handle") is not None:
            # D(x_hat): the generator phase's forward (same weights, same input) -- nothing is launched, the backward runs on its activations
            fake_d = _chain.run_chain_reuse(ctx["d_handle"])[0]
            real_d = (_chain.run_chain_reuse(ctx["d_real_handle"])[0] if ctx.get("d_real_handle") is not None
                      else self.discriminator(ctx["real"], rate_ind=q))
        else:
            # one pass over [x̂; x] (the discriminator has no batch statistics -- norm_type none -- so this equals the
            # reference's two separate calls per sample) halves the launches and doubles the rows of every GEMM
            both = self.discriminator(torch.cat([ctx["fake"], ctx["real"]], dim=0), rate_ind=q)
            fake_d, real_d = both[: both.shape[0] // 2], both[both.shape[0] // 2:]
        l_d_real = self.gan_loss.forward_diff(real_d, fake_d.detach(), is_real=True, is_disc=True) * 0.5
        l_d_fake = self.gan_loss.forward_diff(fake_d, real_d.detach(), is_real=False, is_disc=True) * 0.5
        (l_d_real + l_d_fake).backward()
        self._flush_wgrads("d")
        return {"d_real": l_d_real, "d_fake": l_d_fake, "d_total": l_d_real + l_d_fake,
                "out_d_real": real_d.detach().mean(), "out_d_fake": fake_d.detach().mean()}

    def _d_parts(self, q):
        """The optimiser partition of the sub-discriminator that rate level q trains (None = everything)."""
        return [int(q)] if self.d_optimizer.param_groups[0].get("parts") and len(self.d_optimizer.param_groups[0]["parts"]) > 1 else None

    def _seg_dstep(self, ctx: Dict) -> Dict:
        self.d_optimizer.step(skip=ctx["bad"], partitions=self._d_parts(ctx["q"]))
        return {}

    def optimize_parameters(self, current_iter: int, data_dict: Dict):
        with self._step_scope():
            return self._optimize_parameters(current_iter, data_dict)

    def _optimize_parameters(self, current_iter: int, data_dict: Dict):
        data_dict = dict(data_dict)
        noise = data_dict.pop("noise", None)
        real = self._stage_input(data_dict["real_images"])
        cond, key = self._conditions(data_dict)
        scheduled = bool(getattr(self.rate_loss, "lambda_schedule", None) or getattr(self.rate_loss, "target_rate_schedule", None))
        run = self._runner(key, allow_graph=not scheduled)  # explicit noise tensors must be persistent device buffers
        self.g_optimizer.sync_lr_to_device()
        self.d_optimizer.sync_lr_to_device()
        if self._staged():   # three gradient buckets in backward order, each all-reduce behind the next piece / the D segment
            ctx, g_syncs = self._run_generator_staged(run, real, cond, noise, current_iter)
        else:
            ctx = run("g", lambda: self._seg_generator(real, cond, noise, current_iter))
            g_syncs = [D.AsyncGradSync(self.g_optimizer.flat_grads(), [ctx["bad"]], label="g.all")]       # overlaps the D forward/backward
        ctx_d = run("dfb", lambda: self._seg_dfwdbwd(ctx))
        d_sync = D.AsyncGradSync(self.d_optimizer.flat_grads(partitions=self._d_parts(ctx["q"])), label="d")  # overlaps the G update
        for sy in g_syncs:
            sy.wait()
        ctx2 = run("u", lambda: self._seg_update(ctx))
        d_sync.wait()
        run("d", lambda: self._seg_dstep(ctx))
        log = {"qbpp": ctx["qbpp"] if ctx["qbpp"] is not None else -1, **ctx["losses"], **ctx2, **ctx_d, "_bad": ctx["bad"]}
        return self._finish_step(current_iter, log)
