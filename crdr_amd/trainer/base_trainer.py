"""Training loop skeleton (src/trainer/base_trainer.py:17-238): build models / optimisers / losses / loaders, then
iterate `optimize_parameters`, logging, validation and checkpointing on the reference's step schedule.

Data-parallel extension (the reference is single-GPU): one process per GPU, replicas broadcast from rank 0,
rank 0 alone logs / validates / saves."""
from __future__ import annotations

from typing import Dict, Optional, Union

import torch

from crdr_amd.models import build_comp_model
from crdr_amd.utils.logger import AvgMeter, CSVLogger, bolded_log, get_root_logger, log_dict_items
from crdr_amd.utils.model_saver import Saver
from crdr_amd.utils.path import PathHandler
from crdr_amd.utils.timer import Timer

from . import dist as D


class BaseTrainer:
    def __init__(self, opt) -> None:
        self.opt = opt
        self.device = opt.device
        self.logger = get_root_logger()
        self.is_main = D.rank() == 0
        self.set_models()
        self.set_optimizer_scheduler()
        self.set_losses()
        self.set_dataloader()
        self.use_wandb = False  # wandb is not part of the hot path; rank-0 CSV logs carry the same numbers
        self.loss_recorder = AvgMeter()
        # HIP-graph execution of the step (crdr_amd/trainer/graphs.py): `hip_graphs: true` in the config / CLI overlay
        from .graphs import SegmentGraphs
        self.graphs = SegmentGraphs(bool(opt.get("hip_graphs", False)))
        # weight-gradient reductions of a backward pass are finished by one batched launch (ops.DeferredWgrad); every
        # backward of the trainers is followed by self._flush_wgrads(site)
        self._flush_key = None
        self._deferred = None
        if str(self.device).startswith("cuda") and opt.get("defer_wgrad", True):
            from crdr_amd.hip import ops as _ops
            dev = torch.device(self.device)
            dev = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
            self._deferred = _ops.DeferredWgrad(dev)
        # `precision: bf16x3` (opt-in, default fp32): conv / weight-gradient products of the TRAINING STEP run as split-bf16
        # triples on the bf16 matrix path (include/crdr_hip.h, CRDR_CONV_BF16X3); validation, the codec and all parity claims
        # stay exact fp32.  The reference's own GPU convolutions run TF32 (base_trainer.py:20 + torch 1.12 defaults).
        # `precision: bf16x6` (opt-in): the fp32-EQUIVALENT split (three exact bf16 pieces per operand, six products, CRDR_CONV_BF16X6) in the direct
        # kernels of the training step; held to the fp32 parity gates (tests/test_gpu_bf16x6.py).
        self.precision = str(opt.get("precision", "fp32"))
        if self.precision not in ("fp32", "bf16x3", "bf16x6"):
            raise ValueError(f'precision: "{self.precision}" (fp32, bf16x3 or bf16x6)')
        self.graph_warmup = int(opt.get("hip_graph_warmup", 2))  # eager iterations per graph key before capturing
        self._warm = {}
        self._real_static = None
        self.loss_huge_threshold = 10000.0
        self.time_recorder = Timer(start_iter=opt.get("start_iter", 0), end_iter=opt.get("total_iter", 0))
        if opt.get("path") is not None and self.is_main:
            self.set_csv_loggers()
            self.model_saver = Saver(opt.path.ckpt_root, opt.exp, opt.save_step, opt.keep_step)
        if opt.get("start_iter", 0) > 0:
            self.load_checkpoint(opt.exp, opt.start_iter)
        if opt.get("load_checkpoint", None):
            lc = dict(opt.load_checkpoint)
            self.load_checkpoint(lc.pop("exp"), lc.pop("iter"), **lc)
        if opt.get("dry_run", False):
            self.print_models()
            raise SystemExit(0)

    # ---- construction
    def set_models(self) -> None:
        bolded_log("Model", level="INFO", new_line=True)
        self._set_models()

    def _set_models(self) -> None:
        self.comp_model = build_comp_model(self.opt).to(self.device)
        if self.opt.get("pretrained_weight_path", None):
            self.comp_model.load_learned_weight(self.opt.pretrained_weight_path)
        D.broadcast_module_(self.comp_model)
        cm = getattr(self.comp_model, "context_model", None)
        if cm is not None and hasattr(cm, "seed_noise"):  # in-kernel Philox noise: one stream per rank
            cm.seed_noise((torch.initial_seed() + 7919 * (D.rank() + 1)) & 0x7FFFFFFFFFFFFFFF)
        self.comp_model.train()

    def set_optimizer_scheduler(self) -> None:
        bolded_log("Optimizers & Schedulers", level="INFO", new_line=True)
        self._set_optimizer_scheduler()

    def set_losses(self) -> None:
        bolded_log("Loss functions", level="INFO", new_line=True)
        self._set_losses()

    def set_dataloader(self) -> None:
        """`opt.dataset.train_dataset.type == 'SyntheticDataset'` (or a missing dataset section) yields device-resident
        random crops; otherwise the image-folder datasets of crdr_amd.dataset (reference: base_trainer.py:74-80)."""
        from crdr_amd.dataset import build_loaders
        self.train_loader, self.eval_loader = build_loaders(self.opt, self.device)

    def set_csv_loggers(self) -> None:
        resume = self.opt.get("start_iter", 0) > 0
        self.train_logger = CSVLogger(log_path=self.opt.path.log_loss_path, resume=resume)
        self.eval_logger = CSVLogger(log_path=self.opt.path.log_eval_path, resume=resume)

    def load_checkpoint(self, exp: str, itr: int, **kwargs) -> None:
        bolded_log("Load checkpoint", level="INFO", new_line=True)
        log_dict_items(dict(exp=exp, iter=itr, **kwargs), level="INFO")
        self._load_checkpoint(exp, itr, **kwargs)

    def print_models(self) -> None:
        self.logger.debug(str(self.comp_model))

    # ---- main loop
    def train_data_generator(self, dataloader, start_itr: int, end_itr: int):
        """Endless batches.  Every pass over the loader is a new epoch: a DistributedSampler only reshuffles (and re-deals the
        shards) when told the epoch, which the reference's shuffle=True DataLoader does by itself (base_trainer.py:74-80); a
        resumed run continues the epoch sequence from start_itr."""
        try:
            per_epoch = max(1, len(dataloader))
        except TypeError:
            per_epoch = 1 << 62
        epoch = start_itr // per_epoch

        def fresh():
            sampler = getattr(dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            return iter(dataloader)
        it = fresh()
        for i in range(start_itr, end_itr):
            try:
                data = next(it)
            except StopIteration:
                epoch += 1
                it = fresh()
                data = next(it)
            yield i + 1, data

    def train_loop(self) -> None:
        bolded_log("train_loop start", new_line=True)
        self.time_recorder.start()
        o = self.opt
        for itr, data in self.train_data_generator(self.train_loader, o.get("start_iter", 0), o.total_iter):
            loss_dict = self.optimize_parameters(itr, data)
            if loss_dict is not None:
                self.update_loss_recorder(loss_dict)
            if not self.is_main:
                continue
            if itr % o.log_step == 0:
                self.log_train_loss(itr)
            if self.eval_loader is not None and itr % o.eval_step == 0:
                self.validation(itr)
            if itr % o.save_step == 0:
                self.save(itr)

    def optimize_parameters(self, itr: int, data: Dict) -> Optional[Dict]:
        raise NotImplementedError()

    @staticmethod
    def fetch_scalars(loss_dict: Dict) -> Dict[str, float]:
        """One device->host copy for all logged scalars (the reference does one `.item()` per entry)."""
        keys = list(loss_dict)
        vals = [loss_dict[k] if isinstance(loss_dict[k], torch.Tensor) else torch.tensor(float(loss_dict[k])) for k in keys]
        dev = next((v.device for v in vals if v.is_cuda), torch.device("cpu"))
        flat = torch.stack([v.detach().float().mean().to(dev) for v in vals]).tolist()
        return dict(zip(keys, flat))

    def update_loss_recorder(self, loss_dict: Dict) -> None:
        if any(isinstance(v, torch.Tensor) for v in loss_dict.values()):
            loss_dict = self.fetch_scalars(loss_dict)
        self.loss_recorder.update({k: float(v) for k, v in loss_dict.items()})

    # ---- step plumbing shared by the trainers
    def _stage_input(self, real_images: torch.Tensor) -> torch.Tensor:
        """Copy the batch into a persistent device buffer (NHWC, 4 channel lanes) so captured graphs see a fixed address."""
        from crdr_amd.hip import ops
        x = real_images.to(self.device, non_blocking=True)
        if self._real_static is None or self._real_static.shape != x.shape:
            n, c, h, w = x.shape
            self._real_static = ops.empty_nhwc(n, c, h, w, x.device)
        self._real_static.copy_(x)
        return self._real_static

    def _bad_flag(self, l_total: torch.Tensor) -> torch.Tensor:
        """1.0 if the loss is NaN / Inf / > 1e4 else 0.0 -- the reference's skip-update gate (base_trainer.py:228-238)
        evaluated on the device; the optimiser kernels honour it and the host reads it with the logged scalars."""
        t = l_total.detach().reshape(1)
        return (~torch.isfinite(t) | (t > self.loss_huge_threshold)).float()

    def _flush_wgrads(self, site: str) -> None:
        from crdr_amd.hip import ops as _ops
        # captured and eager executions of a site keep separate job tables: an eager iteration after the capture (e.g. the
        # profiling steps of bench.py) must not rewrite the table a graph replays with
        cap = torch.cuda.is_current_stream_capturing()
        _ops.flush_wgrads((site, self._flush_key, cap), twin=None if cap else (site, self._flush_key, True))

    def _step_scope(self):
        """Context of one optimize_parameters call: the trainer's stream (graphs.step_scope) + deferred weight-gradient
        reductions switched on for exactly this call."""
        return _TrainerStepScope(self)

    def _runner(self, key, allow_graph: bool):
        self._flush_key = key
        """-> callable(name, fn): eager for the first `graph_warmup` iterations of a key (also fills the autotune cache)."""
        n = self._warm.get(key, 0)
        self._warm[key] = n + 1
        if self.graphs.enabled and allow_graph and n >= self.graph_warmup:
            return lambda name, fn: self.graphs.run((name, key), fn)
        return lambda name, fn: fn()

    def validation(self, current_iter: int) -> None:
        self.comp_model.eval()
        df = self.comp_model.validation(self.eval_loader, max_sample_size=100, save_img=False, use_tqdm=False)
        res = df.drop("idx", axis=1).mean().to_dict()
        row = {"iter": current_iter}
        row.update(res)
        self.eval_logger.update(row)
        self.comp_model.train()

    def log_train_loss(self, current_iter: int) -> None:
        row = {"iter": current_iter}
        row.update(self.loss_recorder.get_avg_values())
        self.train_logger.update(row)
        self.loss_recorder.reset()

    def save(self, current_iter: int) -> None:
        raise NotImplementedError()

    # ---- utilities
    @staticmethod
    def update_learning_rate(optimizer, new_lr: float) -> None:
        assert isinstance(new_lr, float)
        for g in optimizer.param_groups:
            g["lr"] = new_lr
        if hasattr(optimizer, "base_lrs"):
            optimizer.base_lrs = [new_lr for _ in optimizer.param_groups]

    @staticmethod
    def check_loss_nan_inf(loss_total) -> Union[str, bool]:
        """'nan' / 'inf' / 'huge' (> 1e4) or False (base_trainer.py:228-238); one host sync."""
        v = float(loss_total.detach().reshape(-1)[0])
        if v != v:
            return "nan"
        if v in (float("inf"), float("-inf")):
            return "inf"
        if v > 10000:
            return "huge"
        return False


class _TrainerStepScope:
    def __init__(self, tr):
        self.tr = tr
        self.inner = tr.graphs.step_scope()

    def __enter__(self):
        from crdr_amd.hip import ops as _ops
        self.prev = _ops.WGRAD_DEFER
        _ops.WGRAD_DEFER = self.tr._deferred
        self.prev_mm = (_ops.MATRIX_BF16X3, _ops.MATRIX_BF16X6)
        _ops.MATRIX_BF16X3, _ops.MATRIX_BF16X6 = self.tr.precision == "bf16x3", self.tr.precision == "bf16x6"
        self.inner.__enter__()
        return self

    def __exit__(self, *a):
        from crdr_amd.hip import ops as _ops
        try:
            return self.inner.__exit__(*a)
        finally:
            if a[0] is not None and _ops.WGRAD_DEFER is not None:
                _ops.WGRAD_DEFER.jobs, _ops.WGRAD_DEFER.off = [], 0  # an exception left reductions behind: drop them
            _ops.WGRAD_DEFER = self.prev
            _ops.MATRIX_BF16X3, _ops.MATRIX_BF16X6 = self.prev_mm
