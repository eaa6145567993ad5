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


class _NoiseState:
    """Checkpoint adapter (Saver.save calls .state_dict()) for the context model's device-resident Philox state."""

    def __init__(self, cm):
        self.cm = cm

    def state_dict(self) -> Dict:
        return self.cm.noise_state()


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
        """Under data parallelism the lambda_A / lambda_B switch sees the global mean of qbpp (eager mode; inside a
        captured graph no collective is possible and the rank-local mean is used -- identical whenever the target
        rates are 0 as in stage 3)."""
        if D.is_dist() and "qbpp" in other and not torch.cuda.is_current_stream_capturing():
            return {"qbpp_mean": D.all_reduce_scalars_mean(other["qbpp"].detach().mean())}
        return {}

    def _generator_update(self, bad: torch.Tensor) -> None:
        """(gradients already all-reduced) global-norm clip folded into the fused Adam."""
        sq = None
        if self.clip_max_norm:
            sq = self.g_optimizer.grad_sqnorm()
            if self.aux_optimizer is not None:  # clip_grad_norm_ runs over comp_model.parameters(): quantiles included
                sq = sq + self.aux_optimizer.grad_sqnorm()
        self.g_optimizer.step(sqnorm=sq, max_norm=self.clip_max_norm, skip=bad)

    def optimize_aux_parameters(self, bad: Optional[torch.Tensor] = None):
        """The quantile gradients depend on the parameters only, so they are identical on every rank: no all-reduce."""
        self.aux_optimizer.zero_grad()
        aux = self.comp_model.aux_loss()
        aux.backward()
        self.aux_optimizer.step(skip=bad)
        return aux

    # ---- segments (each is captured as one HIP graph per condition key)
    def _g_forward(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        """Generator forward + every loss term except the rate loss (which needs the -- possibly global -- mean qbpp)."""
        self.g_optimizer.zero_grad()
        if self.aux_optimizer:
            self.aux_optimizer.zero_grad()
        data = {"real_images": real, **cond}
        if noise is not None:
            data["noise"] = noise
        real_p, fake, bpp, other = self.run_comp_model(data)
        terms = {"distortion": self.distortion_loss(real_p, fake, **other)}
        if self.perceptual_loss:
            terms["perceptual"] = self.perceptual_loss(real_p, fake)
        return {"bpp": bpp, "other": other, "terms": terms, "nonrate": sum(terms.values()), "extra": {}}

    def _seg_generator(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        f = self._g_forward(real, cond, noise, current_iter)
        other = f["other"]
        rate = self.rate_loss(f["bpp"], **other, **self._rate_kwargs(other), current_iter=current_iter)
        l_total = f["nonrate"] + rate
        l_total.backward()
        self._flush_wgrads("g")
        return {"losses": {**f["terms"], "rate": rate}, "bad": self._bad_flag(l_total), "qbpp": other.get("qbpp", None), **f["extra"]}

    # ---- staged generator step (data parallel): forward | all-reduce of mean qbpp | backward in three pieces, each followed by
    # the asynchronous all-reduce of the gradients it completed.  Same arithmetic as _seg_generator: the pieces are disjoint
    # sub-graphs of one backward pass (comp_model.backward_cuts), the rate loss is linear in the two bit sums.
    PIECES = ("decoder", "context_model", None)   # gradient order; None = everything else (hyper-prior, analysis transform)

    def _staged(self) -> bool:
        return D.is_dist() and bool(self.opt.get("dp_buckets", True)) and hasattr(self.comp_model, "backward_cuts") \
            and hasattr(self.comp_model, "context_model")

    def _piece_buffers(self):
        """Flat-gradient slices of each backward piece (parameters are laid out sorted by name, so a piece is one or two
        contiguous runs): [[tensor, ...] per piece]."""
        if getattr(self, "_pieces", None) is None:
            names = {id(p): n for n, p in self.comp_model.named_parameters()}
            pieces = [[] for _ in self.PIECES]
            for g in self.g_optimizer.param_groups:
                if g["grad"] is None:
                    continue
                off, runs = 0, []
                for p in g["params"]:
                    top = names[id(p)].split(".", 1)[0]
                    k = self.PIECES.index(top) if top in self.PIECES else len(self.PIECES) - 1
                    if runs and runs[-1][0] == k and runs[-1][2] == off:
                        runs[-1][2] = off + p.numel()
                    else:
                        runs.append([k, off, off + p.numel()])
                    off += p.numel()
                for k, lo, hi in runs:
                    pieces[k].append(g["grad"][lo:hi])
            self._pieces = pieces
        return self._pieces

    def _seg_gfwd(self, real, cond: Dict, noise, current_iter: int) -> Dict:
        self.comp_model.backward_cuts = {}
        try:
            f = self._g_forward(real, cond, noise, current_iter)
            f["cuts"] = self.comp_model.backward_cuts
        finally:
            self.comp_model.backward_cuts = None
        assert set(f["cuts"]) == {"y", "hyper_out", "y_hat"}, f"staged backward: cuts {sorted(f['cuts'])}"
        q = f["other"].get("qbpp", None)
        f["qbpp_mean"] = q.detach().mean().reshape(1) if q is not None else None   # all-reduced between the segments
        return f

    def _seg_gbwd(self, f: Dict, piece: int, current_iter: int) -> Dict:
        other, cuts = f["other"], f["cuts"]
        if piece == 0:
            kw = {} if f["qbpp_mean"] is None else {"qbpp_mean": f["qbpp_mean"].reshape(())}
            npix = other["num_pixel"]
            rest = {k: v for k, v in other.items() if k != "qbpp_mean"}
            f["rate"] = self.rate_loss(f["bpp"], **rest, **kw, current_iter=current_iter).detach()
            # d rate / d bits is the same constant for both sums: one root per piece
            f["rate_y"] = self.rate_loss(other["bits_y"] / npix, **rest, **kw, current_iter=current_iter)
            f["rate_z"] = self.rate_loss(other["bits_z"] / npix, **rest, **kw, current_iter=current_iter)
            l_total = f["nonrate"].detach() + f["rate"]
            f["nonrate"].backward()
            self._flush_wgrads("g0")
            return {"losses": {**f["terms"], "rate": f["rate"]}, "bad": self._bad_flag(l_total), "qbpp": other.get("qbpp", None), **f["extra"]}
        if piece == 1:
            torch.autograd.backward([cuts["y_hat"][0], f["rate_y"]], [cuts["y_hat"][1].grad, None])
            self._flush_wgrads("g1")
        else:
            torch.autograd.backward([cuts["y"][0], cuts["hyper_out"][0], f["rate_z"]], [cuts["y"][1].grad, cuts["hyper_out"][1].grad, None])
            self._flush_wgrads("g2")
        return {}

    def _run_generator_staged(self, run, real, cond: Dict, noise, current_iter: int):
        """-> (ctx like _seg_generator's, [AsyncGradSync per piece]); the caller waits for the syncs before the update."""
        f = run("gf", lambda: self._seg_gfwd(real, cond, noise, current_iter))
        if f["qbpp_mean"] is not None:   # lambda_A / lambda_B switch on the GLOBAL mean (rate_loss.py:172-175 at global batch)
            D.all_reduce_mean_([f["qbpp_mean"]])
        bufs = self._piece_buffers()
        ctx, syncs = None, []
        for k in range(len(self.PIECES)):
            out = run(f"gb{k}", lambda k=k: self._seg_gbwd(f, k, current_iter))
            ctx = out if k == 0 else ctx
            syncs.append(D.AsyncGradSync(bufs[k], [ctx["bad"]] if k == len(self.PIECES) - 1 else None, label=f"g.{self.PIECES[k] or 'rest'}"))
        return ctx, syncs

    def _seg_update(self, ctx: Dict) -> Dict:
        self._generator_update(ctx["bad"])
        return {"aux": self.optimize_aux_parameters(ctx["bad"])} if self.aux_optimizer else {}

    def _sync_between_segments(self, ctx: Dict, optimizer) -> None:
        D.all_reduce_mean_(optimizer.flat_grads())
        if D.is_dist():
            D.all_reduce_(ctx["bad"], op=torch.distributed.ReduceOp.MAX)

    def _conditions(self, data_dict: Dict) -> Tuple[Dict, object]:
        """-> (model conditioning kwargs, hashable graph key).  A multi-rate model (stage 2) gets its rate index drawn
        here, once per batch like the reference (interpca_hyperprior_model.py:43-45), so that it can key the graph."""
        if hasattr(self.comp_model, "rate_level"):
            q = data_dict.get("rate_ind")
            if q is None:
                if not hasattr(self, "_q_gen"):
                    self._q_gen = torch.Generator().manual_seed(int(self.opt.get("cond_seed", 0)))
                q = torch.randint(self.comp_model.rate_level, (1,), generator=self._q_gen) if D.is_dist() else self.comp_model.sample_rate_ind()
            q = int(q.item()) if isinstance(q, torch.Tensor) else int(q)
            return {"rate_ind": float(q)}, ("rd", q)
        return {}, "rd"

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
        if self._staged():
            ctx, syncs = self._run_generator_staged(run, real, cond, noise, current_iter)
            for sy in syncs:
                sy.wait()
        else:
            ctx = run("g", lambda: self._seg_generator(real, cond, noise, current_iter))
            self._sync_between_segments(ctx, self.g_optimizer)
        ctx2 = run("u", lambda: self._seg_update(ctx))
        log = {"qbpp": ctx["qbpp"] if ctx["qbpp"] is not None else -1, **ctx["losses"], **ctx2, "_bad": ctx["bad"]}
        return self._finish_step(current_iter, log)

    def _finish_step(self, current_iter: int, log: Dict):
        vals = self.fetch_scalars(log)
        if vals.pop("_bad") > 0:
            self.logger.warning(f"iter{current_iter}: skipped because loss is nan / inf / huge (on some rank)")
            return None
        if self.g_scheduler:
            self.g_scheduler.step()
        return vals

    def _training_state(self) -> Dict:
        st = {"g_optimizer": self.g_optimizer}
        if self.aux_optimizer:
            st["aux_optimizer"] = self.aux_optimizer
        if self.g_scheduler:
            st["g_scheduler"] = self.g_scheduler
        cm = getattr(self.comp_model, "context_model", None)
        if cm is not None and hasattr(cm, "noise_state"):   # in-kernel Philox (seed, offset): a resumed run continues it
            st["noise_state"] = _NoiseState(cm)
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
        cm = getattr(self.comp_model, "context_model", None)
        if st.get("noise_state") and cm is not None and hasattr(cm, "load_noise_state"):
            ns = dict(st["noise_state"])   # written by rank 0; every rank keeps a stream of its own
            if ns.get("seed") is not None:
                ns["seed"] = (int(ns["seed"]) + 7919 * D.rank()) & 0x7FFFFFFFFFFFFFFF
            cm.load_noise_state(ns)
