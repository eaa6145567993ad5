"""Optimisers and LR schedulers behind the reference's registry names (`Adam`, `MultiStepLR`;
src/trainer/optimizer/build_optimizer_scheduler.py:11-77).

`Adam` keeps every parameter of a group, its gradient and both moments in ONE flat fp32 buffer each: the
update is a single fused HIP launch (with clip_grad_norm_ folded in as a device-side scale), the gradient
norm is one reduction, and data-parallel training all-reduces one contiguous buffer over RCCL."""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Iterable, List, Optional

import torch

from crdr_amd.hip import functional as HF
from crdr_amd.hip import lib as L
from crdr_amd.hip import ops
from crdr_amd.utils.registry import OPTIMIZER_REGISTRY, SCHEDULER_REGISTRY


class _Group(dict):
    pass


class Adam:
    """torch.optim.Adam semantics (no amsgrad / weight decay) on flat buffers; state_dict() round-trips through the
    torch.optim.Adam format so `training_state_iter*.pth.tar` files stay interchangeable."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False):
        assert weight_decay == 0.0 and not amsgrad, "not used by the CRDR recipes"
        params = list(params)
        groups = params if params and isinstance(params[0], dict) else [{"params": params}]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps)
        self.param_groups: List[Dict] = []
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            grp = _Group(params=ps, lr=g.get("lr", lr), betas=tuple(g.get("betas", betas)), eps=g.get("eps", eps), step=0)
            self._flatten(grp)
            self.param_groups.append(grp)

    @staticmethod
    def _flatten(grp) -> None:
        ps = grp["params"]
        total = sum(p.numel() for p in ps)
        if total == 0:
            grp["flat"] = grp["grad"] = grp["m"] = grp["v"] = None
            return
        dev = ps[0].device
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in ps:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = grad[off:off + n].view(p.shape)
            off += n
        grp["flat"], grp["grad"] = flat, grad
        grp["m"], grp["v"] = torch.zeros_like(flat), torch.zeros_like(flat)

    # ---- partitions: parameter subsets that are updated on their own (torch.optim.Adam skips parameters whose .grad is
    # None -- e.g. the four sub-discriminators a step did not run, module_list_discriminator.py:26-30 -- leaving their
    # moments and step counts untouched; a partition reproduces exactly that on the flat buffers)
    def set_partitions(self, modules: Iterable[torch.nn.Module]) -> None:
        """One partition per module (its parameters must be contiguous in the single param group, in order)."""
        assert len(self.param_groups) == 1, "partitions are defined on a single param group"
        g = self.param_groups[0]
        offs, off = {}, 0
        for p in g["params"]:
            offs[id(p)] = off
            off += p.numel()
        parts, cursor = [], 0
        for m in modules:
            ps = [p for p in m.parameters() if p.requires_grad]
            lo, hi = offs[id(ps[0])], offs[id(ps[-1])] + ps[-1].numel()
            assert lo == cursor and sum(p.numel() for p in ps) == hi - lo, "partition parameters are not contiguous"
            parts.append(_Group(lo=lo, hi=hi, step=g["step"]))
            cursor = hi
        assert cursor == g["flat"].numel(), "partitions must cover the group"
        g["parts"] = parts

    @staticmethod
    def _parts(g, which: Optional[Iterable[int]]):
        if g.get("parts") is None:
            g["parts"] = [_Group(lo=0, hi=g["flat"].numel(), step=g["step"])]
        return g["parts"] if which is None else [g["parts"][i] for i in which]

    # ---- gradient buffers
    def flat_grads(self, partitions: Optional[Iterable[int]] = None) -> List[torch.Tensor]:
        out = []
        for g in self.param_groups:
            if g["grad"] is None:
                continue
            if partitions is None:
                out.append(g["grad"])
            else:
                out += [g["grad"][pt["lo"]:pt["hi"]] for pt in self._parts(g, partitions)]
        return out

    def zero_grad(self, set_to_none: bool = False, partitions: Optional[Iterable[int]] = None) -> None:
        for g in self.param_groups:
            if g["grad"] is not None:
                if partitions is None:
                    g["grad"].zero_()
                else:
                    for pt in self._parts(g, partitions):
                        g["grad"][pt["lo"]:pt["hi"]].zero_()
                off = 0
                for p in g["params"]:  # re-attach in case autograd replaced a .grad tensor
                    n = p.numel()
                    if p.grad is None or p.grad.data_ptr() != g["grad"].data_ptr() + 4 * off:
                        p.grad = g["grad"][off:off + n].view(p.shape)
                    off += n

    def grad_sqnorm(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sum of squared gradients over all groups -> 1-element device tensor (no host sync)."""
        lib = L.load()
        total = None
        for g in self.flat_grads():
            o = torch.empty(1, dtype=torch.float32, device=g.device)
            ws, wsn = ops.workspace(lib.crdr_reduce_workspace(g.numel()), g.device)
            L.check(lib.crdr_sqnorm(g.data_ptr(), g.numel(), o.data_ptr(), ws, wsn, ops._stream()), "sqnorm")
            total = o if total is None else total + o
        if out is not None:
            out.copy_(total)
            return out
        return total

    def _dyn(self, g, pt) -> torch.Tensor:
        """Device-resident [lr, step, skip] of a partition (what the graph-capturable update reads)."""
        if pt.get("dyn") is None:
            pt["dyn"] = torch.tensor([float(g["lr"]), float(pt["step"]), 0.0], dtype=torch.float32, device=g["flat"].device)
            pt["dyn_lr"] = float(g["lr"])
        return pt["dyn"]

    def sync_lr_to_device(self) -> None:
        """Push host-side lr changes (scheduler milestones) into the device scalars; call outside graph capture."""
        for g in self.param_groups:
            for pt in g.get("parts") or []:
                if pt.get("dyn") is not None and pt["dyn_lr"] != float(g["lr"]):
                    pt["dyn"][0:1].fill_(float(g["lr"]))
                    pt["dyn_lr"] = float(g["lr"])

    def step(self, sqnorm: Optional[torch.Tensor] = None, max_norm: float = 0.0, skip: Optional[torch.Tensor] = None,
             partitions: Optional[Iterable[int]] = None) -> None:
        """One Adam update (of the given partitions only). With `sqnorm` (1-element device tensor holding the squared
        global gradient norm) the gradient is scaled by min(1, max_norm / (sqrt(sqnorm) + 1e-6)) inside the kernel
        (clip_grad_norm_).  With `skip` (1-element device tensor, 0 or 1) the step count, lr and the skip decision live
        on the device, so the call sequence is identical every iteration and can be captured in a HIP graph.
        Every weight pack fed by the updated range is refilled by one batched launch right behind the update."""
        lib = L.load()
        if ops.pending_wgrads():
            raise RuntimeError("Adam.step with weight-gradient reductions still pending: call ops.flush_wgrads() after backward")
        touched = []
        for g in self.param_groups:
            if g["flat"] is None:
                continue
            b1, b2 = g["betas"]
            sq = None if sqnorm is None else sqnorm.data_ptr()
            for pt in self._parts(g, partitions):
                lo, n = pt["lo"], pt["hi"] - pt["lo"]
                ptrs = [g[k].data_ptr() + 4 * lo for k in ("flat", "grad", "m", "v")]
                if skip is None:
                    pt["step"] += 1
                    L.check(lib.crdr_adam_step(*ptrs, n, float(g["lr"]), b1, b2, g["eps"], pt["step"], sq, float(max_norm),
                                               ops._stream()), "adam_step")
                else:
                    dyn = self._dyn(g, pt)
                    dyn[1:2].add_(1.0 - skip)   # the step count advances only when the update is applied
                    dyn[2:3].copy_(skip)
                    L.check(lib.crdr_adam_step_dyn(*ptrs, n, b1, b2, g["eps"], dyn.data_ptr(), sq, float(max_norm),
                                                   ops._stream()), "adam_step_dyn")
                touched.append((g, pt))
            g["step"] = max(pt["step"] for pt in g["parts"])
        for g, pt in touched:
            if pt.get("packs") is None:
                pt["packs"] = HF.PackTable(g["flat"], pt["lo"], pt["hi"])
            pt["packs"].refill()

    def host_step_counts(self) -> None:
        """Refresh the host-side step counters from the device (for checkpoints); synchronises."""
        for g in self.param_groups:
            for pt in g.get("parts") or []:
                if pt.get("dyn") is not None:
                    pt["step"] = int(round(float(pt["dyn"][1])))
            if g.get("parts"):
                g["step"] = max(pt["step"] for pt in g["parts"])

    # ---- torch.optim-compatible checkpoint format
    def state_dict(self) -> Dict:
        self.host_step_counts()
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids, off = [], 0
            parts = self._parts(g, None) if g["flat"] is not None else []
            for p in g["params"]:
                n = p.numel()
                step = next(pt["step"] for pt in parts if pt["lo"] <= off < pt["hi"])
                if step > 0:  # torch keeps no state for a parameter that was never stepped
                    state[idx] = {"step": torch.tensor(float(step)), "exp_avg": g["m"][off:off + n].view(p.shape).clone(),
                                  "exp_avg_sq": g["v"][off:off + n].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
                off += n
            groups.append({"lr": g["lr"], "betas": g["betas"], "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: Dict) -> None:
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["betas"], g["eps"] = sg["lr"], tuple(sg["betas"]), sg["eps"]
            if g["flat"] is None:
                continue
            parts = self._parts(g, None)
            for pt in parts:
                pt["step"] = 0
            off = 0
            for p, pid in zip(g["params"], sg["params"]):
                n = p.numel()
                st = sd["state"].get(pid)
                if st is not None:
                    g["m"][off:off + n].copy_(st["exp_avg"].reshape(-1))
                    g["v"][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    pt = next(pt for pt in parts if pt["lo"] <= off < pt["hi"])
                    pt["step"] = max(pt["step"], int(st["step"]))
                off += n
            g["step"] = max(pt["step"] for pt in parts)
            for pt in parts:
                if pt.get("dyn") is not None:
                    pt["dyn"][1:2].fill_(float(pt["step"]))
                    pt["dyn"][0:1].fill_(float(g["lr"]))
                    pt["dyn_lr"] = float(g["lr"])
            HF.bump_weights_epoch()


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR semantics (build_optimizer_scheduler.py:60-77 builds the torch class): the
    schedule is *chainable* -- step() multiplies the optimiser's current lr by gamma when a milestone is reached and
    leaves it alone otherwise, so an lr set from outside (load_checkpoint's new_g_lr / new_d_lr,
    base_trainer.py:update_learning_rate) survives until the next milestone, and load_state_dict restores the counters
    only (the lr itself travels in the optimiser's state dict)."""

    def __init__(self, optimizer, milestones: Iterable[int], gamma: float = 0.1):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0

    def step(self) -> None:
        self.last_epoch += 1
        k = sum(1 for m in self.milestones if m == self.last_epoch)  # a milestone listed twice decays twice
        if k:
            for g in self.optimizer.param_groups:
                g["lr"] = g["lr"] * self.gamma ** k

    def state_dict(self) -> Dict:
        return {"milestones": list(self.milestones), "gamma": self.gamma, "base_lrs": self.base_lrs, "last_epoch": self.last_epoch}

    def load_state_dict(self, sd: Dict) -> None:
        self.last_epoch = sd["last_epoch"]
        self.base_lrs = sd.get("base_lrs", self.base_lrs)
        self.milestones = sorted(sd.get("milestones", self.milestones))
        self.gamma = sd.get("gamma", self.gamma)


OPTIMIZER_REGISTRY.register()(Adam)
SCHEDULER_REGISTRY.register()(MultiStepLR)


def get_params_list(parameters_dict, paramwise_opt, base_lr):
    """Per-key lr multipliers: keys containing any query string form a group with lr = lr_mult * base_lr."""
    groups, remaining = [], dict(parameters_dict)
    for o in paramwise_opt:
        hit = [k for k in sorted(remaining) if any(q in k for q in o["keys"])]
        groups.append({"params": [remaining.pop(k) for k in hit if parameters_dict[k].requires_grad], "lr": o["lr_mult"] * base_lr})
    groups.append({"params": [v for k, v in sorted(remaining.items()) if v.requires_grad]})
    return groups


def build_optimizer(parameters_dict: Dict, optimizer_opt: Dict):
    opt = deepcopy(optimizer_opt)
    opt = opt.to_dict() if hasattr(opt, "to_dict") else dict(opt)
    cls = OPTIMIZER_REGISTRY.get(opt.pop("type"))
    paramwise = opt.pop("paramwise_opt", [])
    params = get_params_list(parameters_dict, paramwise, opt["lr"]) if paramwise else [v for v in parameters_dict.values() if v.requires_grad]
    return cls(params=params, **opt)


def build_scheduler(optimizer, scheduler_opt: Dict):
    opt = deepcopy(scheduler_opt)
    opt = opt.to_dict() if hasattr(opt, "to_dict") else dict(opt)
    return SCHEDULER_REGISTRY.get(opt.pop("type"))(optimizer, **opt)
