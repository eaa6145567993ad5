"""GPU parity of one full training step (stage 3 and stage 1) against the CPU oracle: every loss term, every
generator / discriminator gradient tensor before the optimiser (relative L2 <= 5e-3), the fused Adam + global-norm
clip against torch.optim.Adam + clip_grad_norm_, LPIPS and the GAN/MSE reductions."""
import os

import pytest
import torch

from tests.golden.seeded_weights import seeded_input, seeded_tensor
from tests.test_gpu_model import check_grads, close, dev, grad_sd, rel, seed_module

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(stage: int, bs: int = 2, size: int = 64):
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", f"crdr_stage_{stage}.yaml"))
    cfg.pop("pretrained_weight_path", None)
    cfg["device"] = "cuda:0"
    cfg["dataset"] = {"batch_size": bs, "train_dataset": {"type": "SyntheticDataset", "image_size": size}}
    cfg["path"] = None
    return ConfigDict(cfg)


def _seed_params(module, prefix):
    out = {}
    with torch.no_grad():
        for k, p in module.named_parameters():
            t = seeded_tensor(prefix + k, p.shape)
            p.copy_(t.to(p.device))
            out[prefix + k] = t.clone()
    return out


def test_lpips_and_reductions():
    from oracle import crdr_oracle as O
    from crdr_amd.losses.perceptual_loss import LpipsAlex
    m = LpipsAlex().to(dev())
    sd = _seed_params(m, "lpips.")
    a, b = seeded_input("lp.a", (2, 3, 64, 64)), seeded_input("lp.b", (2, 3, 64, 64))
    bg = b.clone().requires_grad_(True)
    ref = O.lpips_alex(sd, a, bg)
    ref.mean().backward()
    bd = b.to(dev()).requires_grad_(True)
    got = m(a.to(dev()), bd)
    close(got, ref, "lpips value", 2e-4)
    got.mean().backward()
    assert rel(bd.grad, bg.grad) < 2e-3, rel(bd.grad, bg.grad)


def test_adam_matches_torch():
    from crdr_amd.trainer.optimizer import build_optimizer
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s)) for s in ((37, 5), (128,), (3, 3, 3, 3))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for p in ps:
        p.data = p.data.to(dev())
    opt = build_optimizer({str(i): p for i, p in enumerate(ps)}, {"type": "Adam", "lr": 1e-2})
    topt = torch.optim.Adam(ref, lr=1e-2)
    for step in range(4):
        gs = [torch.randn(p.shape) * (10.0 if step == 1 else 0.1) for p in ref]
        opt.zero_grad()
        for p, r, g in zip(ps, ref, gs):
            p.grad.copy_(g.to(dev()))
            r.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        if step % 2 == 0:
            opt.step(sqnorm=opt.grad_sqnorm(), max_norm=1.0)
        else:  # device-side step count / skip flag (the graph-capturable entry): same update
            before = [p.detach().clone() for p in ps]
            opt.step(sqnorm=opt.grad_sqnorm(), max_norm=1.0, skip=torch.ones(1, device=dev()))
            assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps)), "skip flag must leave parameters untouched"
            for g in opt.param_groups:  # host counter mirrors what the eager calls did so far
                for pt in g["parts"]:
                    if pt.get("dyn") is not None:
                        pt["dyn"][1:2].fill_(float(step))
            opt.step(sqnorm=opt.grad_sqnorm(), max_norm=1.0, skip=torch.zeros(1, device=dev()))
            for g in opt.param_groups:
                for pt in g["parts"]:
                    pt["step"] = step + 1
        for p, r in zip(ps, ref):
            close(p, r, f"adam step {step}", 2e-6)
    sd = opt.state_dict()
    topt.load_state_dict(sd)  # format compatibility
    assert len(sd["state"]) == 3


@pytest.mark.parametrize("device_counters", [False, True])
def test_adam_partitions_skip_unused_modules_like_torch(device_counters):
    """torch.optim.Adam leaves parameters whose .grad is None alone (moments, step count); a partitioned flat Adam must
    do the same for the sub-discriminators a step did not run."""
    from crdr_amd.trainer.optimizer import build_optimizer
    torch.manual_seed(3)
    mods = torch.nn.ModuleList(torch.nn.Linear(7, 5) for _ in range(3))
    ref = torch.nn.ModuleList(torch.nn.Linear(7, 5) for _ in range(3))
    ref.load_state_dict(mods.state_dict())
    mods.to(dev())
    opt = build_optimizer(dict(mods.named_parameters()), {"type": "Adam", "lr": 1e-2})
    opt.set_partitions(mods)
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for step, q in enumerate([1, 1, 0, 2, 1, 0]):
        topt.zero_grad(set_to_none=True)
        opt.zero_grad(partitions=[q])
        for p, r in zip(mods[q].parameters(), ref[q].parameters()):
            g = torch.randn(r.shape)
            r.grad = g.clone()
            p.grad.copy_(g.to(dev()))
        topt.step()
        opt.step(partitions=[q], skip=torch.zeros(1, device=dev()) if device_counters else None)
        for (n, p), r in zip(mods.named_parameters(), ref.parameters()):
            close(p, r, f"partitioned adam step {step} {n}", 2e-6)
    sd = opt.state_dict()
    assert [int(sd["state"][i]["step"]) for i in (0, 2, 4)] == [2, 3, 1]
    topt.load_state_dict(sd)


def test_stage3_step():
    _stage3_step()


@pytest.mark.parametrize("q,beta", [(0, 5.12), (4, 1.28)], ids=["q0", "q4"])
def test_stage3_step_over_the_rate_index(q, beta):
    """The trainer's branches on the rate index against the oracle's step (the benchmark cycles all five; q = 2 is test_stage3_step):
    q = 4 takes the REAL image as the relativistic reference (multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:32-34: no high-rate
    pass), q = 0 / 4 train sub-discriminators 0 / 4 with lambda_A[q] and their own Adam partition; the other sub-discriminators must come
    out of the step untouched."""
    _stage3_step(q=q, beta=beta)


def test_stage3_two_iterations_partitioned_d_adam_matches_torch():
    """Two real iterations at q = 1 then q = 3: the discriminator's flat, partitioned Adam against torch.optim.Adam fed the product's own
    captured gradients with `grad = None` for the sub-discriminators a step did not run (module_list_discriminator.py:26-30 +
    torch.optim semantics): sub-D 1 and 3 each moved once with THEIR OWN step count 1 (bias correction of a first step, not of a second),
    sub-Ds 0 / 2 / 4 are bit-identical to their initial values and have no Adam state."""
    from crdr_amd.trainer import build_trainer
    tr = build_trainer(_opt(3, 2, 64))
    _seed_params(tr.comp_model, "")
    sd_d = _seed_params(tr.discriminator, "")
    _seed_params(tr.perceptual_loss.lpips, "lpips.")
    tr.loss_huge_threshold = float("inf")
    ref = {n: torch.nn.Parameter(t.clone()) for n, t in sd_d.items()}
    topt = torch.optim.Adam(list(ref.values()), lr=1e-4)
    d_step = tr.d_optimizer.step
    grads = []

    def wrapped(*a, **k):
        grads.append({n: p.grad.detach().cpu().clone() for n, p in tr.discriminator.named_parameters()})
        return d_step(*a, **k)
    tr.d_optimizer.step = wrapped
    x = seeded_input("image", (2, 3, 64, 64))
    for it, q in enumerate((1, 3), start=1):
        log = tr.optimize_parameters(it, {"real_images": x.to(dev()), "rate_ind": torch.tensor([q]), "beta": 2.56})
        assert log is not None
        topt.zero_grad(set_to_none=True)
        for n, r in ref.items():
            if n.startswith(f"subD_list.{q}."):
                r.grad = grads[-1][n].clone()
                assert float(r.grad.abs().max()) > 0 or n.endswith("16.bias"), n
        topt.step()
    for n, p in tr.discriminator.named_parameters():
        k = int(n.split(".")[1])
        if k in (1, 3):
            close(p, ref[n], f"partitioned D Adam after two iterations: {n}", 2e-6)
            assert not torch.equal(p.detach().cpu(), sd_d[n]) or n.endswith("16.bias"), n
        else:
            assert torch.equal(p.detach().cpu(), sd_d[n]), f"{n}: a sub-discriminator that did not run was touched"
    st = tr.d_optimizer.state_dict()["state"]
    names = [n for n, _ in tr.discriminator.named_parameters()]
    for i, n in enumerate(names):
        k = int(n.split(".")[1])
        step = int(st[i]["step"]) if i in st and "step" in st[i] else 0
        assert step == (1 if k in (1, 3) else 0), (n, step)


def test_stage3_step_winograd():
    """the stage-3 step with every 3x3 stride-1 convolution / input gradient (generator bottlenecks, NLAM, discriminator) on the
    Winograd kernel: same oracle, same gates"""
    from crdr_amd.hip import ops
    ops.PREFER_WINOGRAD = True
    try:
        _stage3_step()
    finally:
        ops.PREFER_WINOGRAD = False


def test_stage3_step_128_winograd_f4x4():
    """the stage-3 step at 128x128 with every convolution the F(4x4, 3x3) Winograd kernel accepts (3x3 stride 1 at >= 48 output columns:
    the 64x64 bottleneck / NLAM stages of both transforms, the discriminator's 128x128 and 64x64 layers and their input gradients) on that
    kernel, the F(2x2, 3x3) kernel on the other 3x3 layers: same oracle, same gates (ops.PREFER_WINOGRAD = 4)"""
    from crdr_amd.hip import ops
    ops.PREFER_WINOGRAD = 4
    try:
        _stage3_step(bs=2, size=128)
    finally:
        ops.PREFER_WINOGRAD = False


# Gradient bound of the parameter groups UPSTREAM of the quantiser (analysis transform, hyper-encoder) under the shipped plan set.
# Their forward activations differ from the oracle's by fp32 summation-order noise; a pre-activation within that noise of zero flips its
# ReLU mask in the backward pass, and every flipped element adds a finite rank-one term to the weight gradients above it.  Behind the
# quantiser the rounding absorbs the noise.  tests/test_conditioning.py reproduces this on the CPU oracle ALONE (float64, noise of the
# kernels' size injected behind the analysis convolutions, shared rounding decisions): 1e-6 of the output scale -> 1.2e-3 upstream,
# 6e-6 -> 3.3e-3, 25 .. 50x less behind the quantiser, 25x less with the clean run's masks imposed -- the figures the GPU measured (1.2e-3 on
# the hyper-encoder with direct / F(2x2) plans; 2.6e-3 .. 9.7e-3 with the F(4x4) kernels at the round-4 points, forward noise ~6e-6;
# profiles/r4_parity_margins.json).  (Round 4 also blamed the heavy-tailed rate gradient for the size of the jumps; the CPU test finds the
# same discrepancy with the rate term off, so that part is withdrawn.)  Round 5's interpolation points bring the F(4x4) kernels' noise to
# ~1.5e-6 (tests/test_gpu_wino.py), hence the tighter cap.
UPSTREAM_TUNED_TOL = 8e-3   # (round 4: 2e-2 for the F(4x4) kernels at the points 0, +-1, +-2; round 5's points carry ~1/4 of that forward noise)
# Round 6: the figure above is a draw from a heavy-tailed distribution (which masks flip depends on the plan set: two re-tunes on the same
# kernels measured 9.4e-3 and 9.8e-3), so a cap on it tested luck and choosing the database that passed was selection on the test.  The gate
# is now DETERMINISTIC: the product exports the ReLU masks its generator's backward used (ops.RELU_MASK_SINK: all 150 sites -- a flip in the
# hyper-synthesis or the context model reaches the hyper-analysis gradients just as one in the analysis transform does), the oracle
# back-propagates through THOSE masks -- adopted only where the pre-activation is within oracle.MASK_WINDOW of zero, at most oracle.MASK_FRACTION of the elements,
# oracle.check_imposed -- exactly as it already adopts the device's rounding decisions, and the upstream gradients are held to the bound
# below for ANY plan set (the shipped database and the rejected re-tune kept under tools/data/).  The un-imposed figure stays a recorded
# margin (tools/parity_margins.sh), not a criterion.
UPSTREAM_IMPOSED_TOL = 5e-4


def _upstream(name: str) -> bool:
    return name.split(".")[0] in ("encoder", "hyperencoder")


N_GENERATOR_RELUS = 150   # analysis 42 (3 x 6 bottleneck + 2 x 12 NLAM) + hyper-analysis 2 + hyper-synthesis 4 + context model 30 x 2 + synthesis 42


class _MaskSink:
    """collects, while active, the ReLU masks the product's generator saved for its backward, keyed by the oracle's conv names"""

    def __init__(self, model):
        self.names = {p.data_ptr(): n[:-len(".weight")] for n, p in model.named_parameters() if n.endswith(".weight")}
        self.masks = {}

    def __call__(self, weight, out, offset):
        n = self.names.get(weight.data_ptr())
        if n is not None and n not in self.masks:
            a = out.detach()
            if offset is not None:   # a beta vector was added after the ReLU: the backward compares act - vector (CRDR_EPI_MASKOFF), in fp32
                a = a - offset.detach().reshape(1, -1, 1, 1)
            self.masks[n] = (a > 0).cpu()

    def __enter__(self):
        from crdr_amd.hip import ops
        ops.RELU_MASK_SINK = self
        return self

    def __exit__(self, *exc):
        from crdr_amd.hip import ops
        ops.RELU_MASK_SINK = None


REJECTED_RETUNE = os.path.join(ROOT, "tools", "data", "tune_r5_rejected_c.json")   # the round-5 candidate that measured 9.8e-3 un-imposed


class _ShippedPlans:
    """what bench.py and scripts/train.py run: ops.AUTOTUNE on with the shipped perf database (crdr_amd/hip/tune_gfx950.json), or another
    plan set (`db`: a database kept under tools/data/)"""

    def __init__(self, db=None):
        self.db = db

    def __enter__(self):
        from crdr_amd.hip import ops
        self.ops, self.keep = ops, (ops.AUTOTUNE, dict(ops._algo_cache))
        ops._algo_cache.clear()
        if self.db is not None:
            assert ops.load_tune_cache(self.db, ignore_signature=True) > 0, self.db
        else:
            assert ops.load_tune_cache(ops.DEFAULT_TUNE_DB) > 0, "the shipped perf database is not of this library build"
        ops.AUTOTUNE = True
        self.log0 = len(ops.TUNE_LOG)
        return self

    def tuned_here(self):
        """shapes the database did not hold (tuned on the spot, with the agreement check of ops._autotune)"""
        return [k for k, *_ in self.ops.TUNE_LOG[self.log0:]]

    def __exit__(self, *exc):
        self.ops.AUTOTUNE = self.keep[0]
        self.ops._algo_cache.clear()
        self.ops._algo_cache.update(self.keep[1])


def test_stage3_step_256_tuned_vs_oracle():
    """BASELINE config #3 as benchmarked: bs 16, 256x256 crops, the shipped per-shape plans (tile / split / streaming 1x1 /
    Winograd ids) -- every loss term and every G / D / aux gradient tensor against the oracle's step
    (multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:13-114), same forced-decision windows as the 64x64 test."""
    from tests import parity_margins as PM
    with _ShippedPlans() as sp:
        _stage3_step(bs=16, size=256, impose_masks=True)
        new = sp.tuned_here()
    PM.record("plans", "shapes tuned on the spot (not in the shipped database)", float(len(new)))
    assert len(new) <= 8, new


def test_stage1_step_256_tuned_vs_oracle():
    """BASELINE config #2 (bs 8, 256x256) under the shipped plans against the oracle's stage-1 step (rate_distortion_trainer.py:57-101)"""
    from tests import parity_margins as PM
    with _ShippedPlans() as sp:
        _stage1_step(bs=8, size=256, impose_masks=True)
        new = sp.tuned_here()
    PM.record("plans", "shapes tuned on the spot (not in the shipped database)", float(len(new)))
    assert len(new) <= 8, new


def test_stage1_step_256_rejected_retune_passes_the_deterministic_gate():
    """The plan set round 5 REJECTED (a re-tune on the same kernels whose un-imposed upstream figure was 9.8e-3 against the 8e-3 cap: 121 of
    759 entries differ from the shipped set, a dozen small hyper-path layers on the F(4x4) kernels) under the deterministic gate: with the
    device's ReLU masks imposed its upstream gradients are as close to the oracle's as the shipped set's -- the old failure was the mask
    lottery, not a kernel.  (If this ever fails with the masks imposed, that IS a kernel bug.)"""
    with _ShippedPlans(REJECTED_RETUNE):
        _stage1_step(bs=8, size=256, impose_masks=True)


def _stage3_step(bs: int = 2, size: int = 64, upstream_tol: float = 0.0, q: int = 2, beta: float = 2.56, precision: str = "fp32",
                 impose_masks: bool = False):
    from oracle import crdr_oracle as O
    from crdr_amd.trainer import build_trainer
    opt = _opt(3, bs, size)
    opt["precision"] = precision
    tr = build_trainer(opt)
    # the optimisers flattened the parameters on the device: seed in place (views are preserved)
    sd_g = _seed_params(tr.comp_model, "")
    sd_d = _seed_params(tr.discriminator, "")
    sd_l = _seed_params(tr.perceptual_loss.lpips, "lpips.")
    x = seeded_input("image", (bs, 3, size, size))
    ny = seeded_input("noise.y", (bs, 320, size // 16, size // 16), 0.5)
    nz = seeded_input("noise.z", (bs, 192, size // 64, size // 64), 0.5)
    has_hr = q + 1 <= 4   # (the top rate compares against the real image: no high-rate pass)

    captured = {}
    g_step, d_step, a_step = tr.g_optimizer.step, tr.d_optimizer.step, tr.aux_optimizer.step

    def cap(name, module, fn):
        def wrapped(*a, **k):
            captured[name] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in module.named_parameters()}
            return fn(*a, **k)
        return wrapped
    tr.g_optimizer.step = cap("g", tr.comp_model, g_step)
    tr.d_optimizer.step = cap("d", tr.discriminator, d_step)
    tr.aux_optimizer.step = cap("aux", tr.comp_model, a_step)
    before = {n: p.detach().clone() for n, p in tr.comp_model.named_parameters()}

    data = {"real_images": x.to(dev()), "rate_ind": torch.tensor([q]), "beta": beta,
            "noise": {"y": ny.to(dev()), "z": nz.to(dev())}}
    tr.loss_huge_threshold = float("inf")  # seeded random weights give a loss > 1e4, which the trainer would skip
    tr.comp_model.context_model.record_symbols = []
    z_hats = []
    run_model = tr.comp_model.run_model

    def spy(*a, **k):
        o = run_model(*a, **k)
        z_hats.append(o["z_hat"].detach().cpu())
        return o
    tr.comp_model.run_model = spy
    reconstruct = tr.comp_model.reconstruct

    def spy_hr(*a, **k):  # the no-grad high-rate pass
        o = reconstruct(*a, **k)
        z_hats.append(o["z_hat"].detach().cpu())
        return o
    tr.comp_model.reconstruct = spy_hr
    with _MaskSink(tr.comp_model) as sink:
        log = tr.optimize_parameters(1, data)
    assert log is not None
    impose = {"masks": sink.masks, "report": {}} if impose_masks else None
    if impose_masks:
        assert len(sink.masks) == N_GENERATOR_RELUS, sorted(sink.masks)
        upstream_tol = UPSTREAM_IMPOSED_TOL
    syms = [t.cpu() for t in tr.comp_model.context_model.record_symbols]
    assert len(syms) == (20 if has_hr else 10) and len(z_hats) == (2 if has_hr else 1)
    med = sd_g["entropy_model_z.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    forced = {"y": syms[:10], "z": torch.round(z_hats[0] - med)}
    hr_forced = {"y": syms[10:], "z": torch.round(z_hats[1] - med)} if has_hr else None
    g_ref, d_ref, rep = grad_sd(sd_g), grad_sd(sd_d), {}
    losses, out = O.stage3_g_losses(g_ref, d_ref, sd_l, x, q, beta, ny, nz, forced=forced, hr_forced=hr_forced, report=rep, impose=impose)
    O.check_forced(rep, rep.get("symbols", 0))
    if impose is not None:
        O.check_imposed(impose["report"])
        from tests import parity_margins as PM_
        PM_.record_imposed(impose["report"])
    losses["total"].backward()
    d_ref = grad_sd(sd_g), grad_sd(sd_d)
    d_ref = d_ref[1]
    d_losses = O.stage3_d_losses(d_ref, x, out["fake_images"], q)
    d_losses["d_total"].backward()
    aux_ref = grad_sd(sd_g)
    O.eb_aux_loss(aux_ref, "entropy_model_z").backward()
    for k in ("distortion", "rate", "perceptual", "adv"):
        close(log[k], losses[k], f"loss {k}", 3e-4)
    for k in ("d_real", "d_fake"):
        close(log[k], d_losses[k], f"loss {k}", 3e-4)
    close(log["aux"], O.eb_aux_loss(sd_g, "entropy_model_z"), "aux loss", 1e-4)
    close(log["qbpp"], out["qbpp"].mean(), "qbpp", 1e-4)

    from tests import parity_margins as PM

    def cmp(cap_d, ref_sd, what, only=None, tol=5e-3, upstream_tol=upstream_tol):
        bad = []
        for n, g in cap_d.items():
            if only is not None and not only(n):
                continue
            r = ref_sd[n].grad
            if r is None or r.abs().max() <= 1e-7:
                # analytically zero (e.g. the D head bias cancels in D(x) - D(x̂)): what either side holds is the rounding residue of its own
                # summation order -- exactly 0 or a few 2^-25 .. 2^-29 (a re-tuned plan set turned the product's -3.0e-8 into -1.9e-9 against the
                # oracle's -3.0e-8: "relative error 0.94" of nothing) -- so both are held to the same absolute floor
                assert g is None or g.abs().max().item() <= 1e-7, f"{what}: unexpected gradient for {n}"
                continue
            e = rel(g, r)
            grp = f"grad:{what}:" + PM.group_of(n)
            PM.record(grp, n, e)
            t = PM.tolerance(grp, upstream_tol if (upstream_tol and _upstream(n)) else tol)
            if e > t:
                bad.append((n, e, t) + ((g.flatten().tolist(), r.flatten().tolist()) if g.numel() <= 4 else ()))
        assert not bad, f"{what}: {bad[:8]} ({len(bad)})"
    cmp(captured["g"], g_ref, "G grads", only=lambda n: not n.endswith(".quantiles"))
    cmp(captured["d"], d_ref, "D grads")
    ran = [n for n, r in d_ref.items() if r.grad is not None and float(r.grad.abs().max()) > 1e-7]
    assert ran and all(n.startswith(f"subD_list.{q}.") for n in ran), ran[:3]
    cmp(captured["aux"], aux_ref, "aux grads", only=lambda n: n.endswith(".quantiles"))
    # the generator update really happened, with the clipped step size bounded by lr
    moved = max((p.detach() - before[n]).abs().max().item() for n, p in tr.comp_model.named_parameters())
    assert 0 < moved <= 1.0e-3 + 1e-6


def test_stage1_step():
    _stage1_step()


def _stage1_step(bs: int = 2, size: int = 64, upstream_tol: float = 0.0, precision: str = "fp32", impose_masks: bool = False):
    from oracle import crdr_oracle as O
    from crdr_amd.trainer import build_trainer
    opt = _opt(1, bs, size)
    opt["precision"] = precision
    tr = build_trainer(opt)
    sd_g = _seed_params(tr.comp_model, "")
    sd_l = _seed_params(tr.perceptual_loss.lpips, "lpips.")
    x = seeded_input("image", (bs, 3, size, size))
    ny = seeded_input("noise.y", (bs, 320, size // 16, size // 16), 0.5)
    nz = seeded_input("noise.z", (bs, 192, size // 64, size // 64), 0.5)
    captured = {}
    g_step = tr.g_optimizer.step
    tr.loss_huge_threshold = float("inf")
    tr.comp_model.context_model.record_symbols = []
    z_hats = []
    run_model = tr.comp_model.run_model

    def spy(*a, **k):
        o = run_model(*a, **k)
        z_hats.append(o["z_hat"].detach().cpu())
        return o
    tr.comp_model.run_model = spy

    def wrapped(*a, **k):
        captured.update({n: (p.grad.clone() if p.grad is not None else None) for n, p in tr.comp_model.named_parameters()})
        return g_step(*a, **k)
    tr.g_optimizer.step = wrapped
    with _MaskSink(tr.comp_model) as sink:
        log = tr.optimize_parameters(1, {"real_images": x.to(dev()), "noise": {"y": ny.to(dev()), "z": nz.to(dev())}})
    impose = {"masks": sink.masks, "report": {}} if impose_masks else None
    if impose_masks:
        assert len(sink.masks) == N_GENERATOR_RELUS, sorted(sink.masks)
        upstream_tol = UPSTREAM_IMPOSED_TOL
    med = sd_g["entropy_model_z.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    forced = {"y": [t.cpu() for t in tr.comp_model.context_model.record_symbols], "z": torch.round(z_hats[0] - med)}
    g_ref, rep = grad_sd(sd_g), {}
    losses, out = O.stage1_losses(g_ref, sd_l, x, ny, nz, forced=forced, report=rep, impose=impose)
    O.check_forced(rep, rep.get("symbols", 0))
    if impose is not None:
        O.check_imposed(impose["report"])
        from tests import parity_margins as PM_
        PM_.record_imposed(impose["report"])
    losses["total"].backward()
    for k in ("distortion", "rate", "perceptual"):
        close(log[k], losses[k], f"loss {k}", 3e-4)
    from tests import parity_margins as PM
    bad = []
    for n, g in captured.items():
        r = g_ref[n].grad
        if n.endswith(".quantiles") or r is None or r.abs().max() == 0:
            continue
        e = rel(g, r)
        grp = "grad:G grads:" + PM.group_of(n)
        PM.record(grp, n, e)
        t = PM.tolerance(grp, upstream_tol if (upstream_tol and _upstream(n)) else 5e-3)
        if e > t:
            bad.append((n, e, t))
    assert not bad, bad[:8]


def test_stage1_training_reduces_the_loss():
    """Not a parity statement, a liveness one: 150 stage-1 iterations on one fixed batch (graph-replayed) must drive the
    R-D objective down by a wide margin -- gradients, clipping, the fused Adam, the batched weight-pack refill and the
    deferred weight-gradient reduction all have to cooperate for that."""
    import bench
    tr = bench.build_trainer(1, 4, 64, "cuda:0", graphs=True)
    x = (torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1).to("cuda:0")
    x = torch.nn.functional.avg_pool2d(x, 4).repeat_interleave(4, 2).repeat_interleave(4, 3)  # blocky, learnable
    first, last = [], []
    for it in range(1, 151):
        log = tr.optimize_parameters(it, {"real_images": x})
        assert log is not None, f"iteration {it} was skipped"
        tot = log["distortion"] + log["rate"] + log["perceptual"]
        (first if it <= 5 else last if it > 145 else []).append(tot)
    a, b = sum(first) / len(first), sum(last) / len(last)
    assert b < 0.6 * a, (a, b)
