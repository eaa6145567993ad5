"""BASELINE config #4 (stage-3 data parallel, RCCL all-reduce of G + D gradients) on ONE GPU:

* the real trainer's data-parallel path through a 1-rank RCCL group (`CRDR_FORCE_DIST=1`, child processes, HIP graphs on
  and off) logs the same scalars and ends with bit-identical generator / discriminator parameters as the plain run;
* the data-parallel identity on the real trainer: the mean of the flat G / D gradient buffers of two half-batch passes
  equals the full-batch pass (what the all-reduce computes across ranks), with the skip flag OR-ed.
The 2-rank collective logic itself runs on CPU in tests/test_dp_gloo.py (gloo)."""
import os
import subprocess
import sys

import pytest
import torch

from tests.golden.seeded_weights import seeded_input
from tests.test_gpu_model import dev, rel
from tests.test_gpu_step import _opt, _seed_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_batch(tmp, force_dist, variants):
    """ONE child process for all `variants` ({name: [worker options]}) of one kind -- plain, or on the 1-rank RCCL process group
    (CRDR_FORCE_DIST=1) -- each with a fresh trainer and freshly seeded generators (tests/dp_step_worker.py --batch); -> {name: dump}"""
    import json
    env = dict(os.environ, CRDR_FORCE_DIST="1" if force_dist else "0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    tag = "dp" if force_dist else "plain"
    spec = [{"out": str(tmp / f"{tag}_{name}.pt"), "args": list(args)} for name, args in variants.items()]
    spec_path = tmp / f"{tag}_spec.json"
    spec_path.write_text(json.dumps(spec))
    r = subprocess.run([sys.executable, "-m", "tests.dp_step_worker", "--batch", str(spec_path)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return {name: torch.load(sp["out"], map_location="cpu", weights_only=False) for name, sp in zip(variants, spec)}


STRADDLE = ("--stage", "3", "--fixed-q", "2", "--global-bs", "4", "--forced-algo", "1", "--graphs", "--shard", "--straddle-target")


@pytest.fixture(scope="module")
def dp_runs(tmp_path_factory):
    """Every child-process variant of this file, in TWO processes (round 4: eighteen, 212 s of the suite): the plain runs in one, the
    runs on the 1-rank RCCL group in the other.  Keys: (kind, name)."""
    tmp = tmp_path_factory.mktemp("dp_runs")
    staged = {"stage1": ("--stage", "1"), "stage3-lambdaB": ("--stage", "3", "--target-rate", "100.0"),
              "stage3-target0.5": ("--stage", "3", "--target-rate", "0.5")}
    plain = {"graphs": ["--graphs"], "eager": [], "straddle": list(STRADDLE)}
    dp = {"graphs": ["--graphs"], "eager": [], "draw_a": ["--draw-conditions", "--iters", "4"], "draw_b": ["--draw-conditions", "--iters", "4"]}
    for name, extra in staged.items():
        plain["staged_" + name] = ["--graphs", *extra]
        dp["staged_" + name] = ["--graphs", *extra]
        dp["single_" + name] = ["--graphs", *extra, "--no-buckets"]
    out = {("plain", k): v for k, v in _run_batch(tmp, False, plain).items()}
    out.update({("dp", k): v for k, v in _run_batch(tmp, True, dp).items()})
    return out


@pytest.mark.parametrize("graphs", [True, False])
def test_trainer_dp_path_on_one_rank_rccl_is_bit_identical(dp_runs, graphs):
    plain, dp = dp_runs[("plain", "graphs" if graphs else "eager")], dp_runs[("dp", "graphs" if graphs else "eager")]
    assert plain["dist"] is False and dp["dist"] is True and dp["world"] == 1
    if graphs:
        assert dp["graphs"] >= 4, "the data-parallel run did not capture its segments"
    for a, b in zip(plain["logs"], dp["logs"]):
        assert a is not None and b is not None and a.keys() == b.keys()
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])
    for part in ("G", "D"):
        for k in plain[part]:
            assert torch.equal(plain[part][k], dp[part][k]), (part, k)


@pytest.mark.parametrize("name", ["stage1", "stage3-lambdaB", "stage3-target0.5"])
def test_staged_backward_and_global_rate_switch_are_bit_identical(dp_runs, name):
    """Config #4 semantics on one rank, HIP graphs on: the data-parallel generator step -- forward graph | all-reduce of the
    mean qbpp (lambda_A / lambda_B on the GLOBAL mean, rate_loss.py:172-175) | backward in three captured pieces with one
    asynchronous gradient bucket each -- against (a) the same path with one bucket behind an unstaged backward and (b) the
    plain single-GPU run: same logs, bit-identical parameters.  target 100 forces the lambda_B branch, 0.5 sits among the
    seeded model's qbpp values; stage 1 (crdr_stage_1.yaml: HificRateLoss, target 1.5) is the path where the all-reduce is
    otherwise fully exposed."""
    plain, staged, single = dp_runs[("plain", "staged_" + name)], dp_runs[("dp", "staged_" + name)], dp_runs[("dp", "single_" + name)]
    assert staged["staged"] is True and single["staged"] is False and plain["staged"] is False
    assert staged["graphs"] >= single["graphs"] + 3, (staged["graphs"], single["graphs"])
    for other in (single, plain):
        for a, b in zip(other["logs"], staged["logs"]):
            assert a is not None and b is not None and a.keys() == b.keys()
            for k in a:
                assert a[k] == b[k], (k, a[k], b[k])
        for part in ("G", "D"):
            for k in other[part]:
                assert torch.equal(other[part][k], staged[part][k]), (part, k)


def test_dp_ranks_draw_shared_conditions(dp_runs):
    """with the trainer drawing (q, beta) itself, the data-parallel path uses the seeded shared generators
    (same sequence on every rank): two runs (fresh trainers) give identical logs"""
    a, b = dp_runs[("dp", "draw_a")], dp_runs[("dp", "draw_b")]
    assert [l["qbpp"] for l in a["logs"]] == [l["qbpp"] for l in b["logs"]]
    assert len({round(l["qbpp"], 6) for l in a["logs"]}) > 1, "the rate index never changed over 4 draws"


def test_dp_identity_on_the_real_trainer():
    """mean over two half-batch 'ranks' of the flat gradient buffers == the full-batch gradients (generator and active
    sub-discriminator), same (q, beta), explicit per-sample noise.

    Tolerances.  Everything downstream of the quantiser is bit-identical in the forward pass between the batch splits
    (asserted on x_hat: rounding absorbs the fp32 summation-order noise of the analysis transform, whose tile / split-K
    choice depends on the batch size), and its gradients -- decoder, LRP transforms, hyper-decoder / -encoder, the active
    sub-discriminator -- agree to 2e-5 relative L2.  The ENCODER's own forward differs in the last bits between the splits;
    a pre-activation that lands on the other side of zero flips its ReLU mask in the backward (a finite jump: measured on
    the encoder alone, 3e-5 median / 5e-4 worst per tensor at these sizes), so the encoder is held to
    5e-3; the mean / scale transforms see y only through the heavy-tailed -1/(p ln 2) likelihood gradient (p down to the
    1e-9 floor under seeded random weights) and are held to 5e-4 with the rate term on, and are exactly zero with it off.

    The library picks a tile configuration / split-K depth per problem size, and a different split means a different fp32
    summation order; data-parallel ranks all run the same per-rank batch, so for them the choice coincides.  Here the "ranks"
    have half the batch of the reference run, hence the convolutions are pinned to one unsplit configuration
    (ops.FORCED_CONV_ALGO) for the duration of the test."""
    from crdr_amd.hip import ops
    from crdr_amd.trainer import build_trainer
    ops.FORCED_CONV_ALGO = 1
    try:
        _dp_identity_body(build_trainer)
    finally:
        ops.FORCED_CONV_ALGO = 0


def _dp_identity_body(build_trainer):
    tr = build_trainer(_opt(3, bs=4))
    _seed_params(tr.comp_model, "")
    _seed_params(tr.discriminator, "")
    _seed_params(tr.perceptual_loss.lpips, "lpips.")
    tr.loss_huge_threshold = float("inf")
    x = seeded_input("image4", (4, 3, 64, 64)).to(dev())
    ny = seeded_input("noise4.y", (4, 320, 4, 4), 0.5).to(dev())
    nz = seeded_input("noise4.z", (4, 192, 1, 1), 0.5).to(dev())
    q, beta = 1, 3.2
    names = {id(p): n for n, p in tr.comp_model.named_parameters()}
    g_params = [p for g in tr.g_optimizer.param_groups for p in g["params"]]

    def grads(sl):
        with tr._step_scope():
            cond, key = tr._conditions({"rate_ind": q, "beta": beta})
            tr._runner(key, allow_graph=False)
            real = tr._stage_input(x[sl])
            ctx = tr._seg_generator(real, cond, {"y": ny[sl], "z": nz[sl]}, 1)
            tr._seg_dfwdbwd(ctx)
            g = torch.cat([b.reshape(-1) for b in tr.g_optimizer.flat_grads()]).clone()
            d = torch.cat([b.reshape(-1) for b in tr.d_optimizer.flat_grads(partitions=tr._d_parts(q))]).clone()
            return g, d, float(ctx["bad"]), {k: float(v.detach()) for k, v in ctx["losses"].items()}, ctx["fake"].clone()

    def by_module(avg, full):
        off, acc = 0, {}
        for p in g_params:
            n = p.numel()
            top = names[id(p)].split(".")
            top = ".".join(top[:2]) if top[0] == "context_model" else top[0]
            a = acc.setdefault(top, [0.0, 0.0])
            a[0] += float((avg[off:off + n] - full[off:off + n]).double().square().sum())
            a[1] += float(full[off:off + n].double().square().sum())
            off += n
        assert off == full.numel()
        return {k: (d2 / max(n2, 1e-300)) ** 0.5 for k, (d2, n2) in acc.items()}

    lam = (list(tr.rate_loss.lambda_A), list(tr.rate_loss.lambda_B))
    for rate_on in (False, True):
        if not rate_on:
            tr.rate_loss.lambda_A, tr.rate_loss.lambda_B = [0.0] * len(lam[0]), [0.0] * len(lam[1])
        else:
            tr.rate_loss.lambda_A, tr.rate_loss.lambda_B = lam
        gf, df, badf, lf, ff = grads(slice(0, 4))
        g0, d0, bad0, l0, f0 = grads(slice(0, 2))
        g1, d1, bad1, l1, f1 = grads(slice(2, 4))
        assert torch.equal(torch.cat([f0, f1]), ff), "the forward must not depend on how the batch is split"
        assert gf.abs().max() > 0 and df.abs().max() > 0
        assert rel((d0 + d1) / 2, df) < 2e-5
        errs = by_module((g0 + g1) / 2, gf)
        tol = {"encoder": 5e-3, "context_model.mean_slice_transforms": 5e-4 if rate_on else 1e-30,
               "context_model.scale_slice_transforms": 5e-4 if rate_on else 1e-30}
        for k, e in errs.items():
            assert e < tol.get(k, 2e-5), (rate_on, k, e, errs)
        assert max(bad0, bad1) == badf == 0.0
        for k in lf:
            assert abs((l0[k] + l1[k]) / 2 - lf[k]) <= 2e-5 * max(1.0, abs(lf[k])), (k, l0[k], l1[k], lf[k])


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE config #4 on real ranks: W processes under torch.distributed.run, one GPU each, RCCL.  W = 1 runs everywhere (the
# launcher, per-rank device selection, sharding and dump plumbing on a 1-rank RCCL group); W = 2 runs the moment two GPUs are
# visible and is the first N > 1 execution of the staged data-parallel step.
# ---------------------------------------------------------------------------------------------------------------------------
def _launch_ranks(out, world, extra, port):
    """fresh child processes (never an exec from this GPU-initialised one): torchrun -> tests.dp_step_worker on cuda:LOCAL_RANK"""
    env = dict(os.environ, CRDR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "tests.dp_step_worker", str(out), "--graphs", "--shard"] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return [torch.load(f"{out}.rank{k}", map_location="cpu", weights_only=False) for k in range(world)]


def _by_module(layout, got, ref):
    off, acc = 0, {}
    for name, n in layout:
        top = name.split(".")
        top = ".".join(top[:2]) if top[0] == "context_model" else top[0]
        a = acc.setdefault(top, [0.0, 0.0])
        a[0] += float((got[off:off + n] - ref[off:off + n]).double().square().sum())
        a[1] += float(ref[off:off + n].double().square().sum())
        off += n
    assert off == ref.numel() == got.numel()
    return {k: (d2 / max(n2, 1e-300)) ** 0.5 for k, (d2, n2) in acc.items()}


@pytest.mark.parametrize("world", [1, 2])
def test_two_rank_rccl_staged_step(tmp_path, dp_runs, world):
    """Stage 3, HIP graphs on, staged generator step (forward graph | all-reduce of the mean qbpp | three backward graphs with one
    asynchronous RCCL bucket each), a global batch of 4 seeded images with explicit noise split over `world` ranks:

    * the lambda_A / lambda_B switch is STRADDLED: the target rate sits between the two half-batch means of the quantised bpp (every
      process derives it from the seeded parameters before it trains, dp_step_worker --straddle-target), so a rank deciding on its
      local mean would take the other branch than the global mean does (rate_loss.py:172-175 at global batch);
    * all ranks hold bit-identical generator and discriminator parameters after 5 iterations;
    * the all-reduced gradients of iteration 1 equal the one-GPU run of the whole batch within the data-parallel identity's
      tolerances (test_dp_identity_on_the_real_trainer: 2e-5 behind the quantiser, encoder 5e-3, mean / scale transforms 5e-4) --
      a rank on the wrong lambda branch would be off by the ratio lambda_A / lambda_B in the rate gradient."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (have {torch.cuda.device_count()})")
    gbs = 4
    # the whole batch on one GPU, plain (no process group): the reference the ranks must reproduce
    ref = dp_runs[("plain", "straddle")]
    assert ref["dist"] is False
    qb, target = ref["probe_qbpp"], ref["target"]
    lo, hi = sorted([float(qb[: gbs // 2].mean()), float(qb[gbs // 2:].mean())])
    assert hi - lo > 1e-4 * hi, ("the two halves of the seeded batch have the same mean qbpp", lo, hi)
    glob = float(qb.mean())
    assert lo < target < glob < hi      # global decision lambda_A, the low half alone would say lambda_B
    ranks = _launch_ranks(tmp_path / "dp.pt", world, [x for x in STRADDLE if x not in ("--graphs", "--shard")], 29650 + world)
    assert all(s["dist"] and s["world"] == world and s["staged"] for s in ranks)
    assert all(s["target"] == target for s in ranks), ([s["target"] for s in ranks], target)
    assert all(s["graphs"] >= 7 for s in ranks), [s["graphs"] for s in ranks]
    assert all(lg is not None for s in ranks for lg in s["logs"]), "an iteration was skipped"
    if world == 2:   # the straddle is real on the ranks' own logs (iteration 1: the probe's parameters)
        l0, l1 = sorted(float(s["logs"][0]["qbpp"]) for s in ranks)
        assert l0 < target < l1, (l0, target, l1)
    for s in ranks[1:]:
        for part in ("G", "D"):
            for k in ranks[0][part]:
                assert torch.equal(ranks[0][part][k], s[part][k]), (part, k)
        assert torch.equal(ranks[0]["grads"]["G"], s["grads"]["G"]) and torch.equal(ranks[0]["grads"]["D"], s["grads"]["D"])
    errs = _by_module(ranks[0]["g_layout"], ranks[0]["grads"]["G"], ref["grads"]["G"])
    if world == 1:     # same batch, same plans: the 1-rank RCCL path changes nothing
        assert max(errs.values()) == 0.0, errs
        assert torch.equal(ranks[0]["grads"]["D"], ref["grads"]["D"])
        for part in ("G", "D"):
            for k in ref[part]:
                assert torch.equal(ranks[0][part][k], ref[part][k]), (part, k)
        return
    tol = {"encoder": 5e-3, "context_model.mean_slice_transforms": 5e-4, "context_model.scale_slice_transforms": 5e-4}
    for k, e in errs.items():
        assert e < tol.get(k, 2e-5), (k, e, errs)
    assert rel(ranks[0]["grads"]["D"], ref["grads"]["D"]) < 2e-5


def test_bench_two_gpus_reports_two_rccl_ranks():
    """`python bench.py --gpus 2 --steps 5` (the driver's scaling command, self-launched) on two GPUs: the line says n_gpus 2, the
    process group really had two RCCL ranks (all-reduce of ones = 2) and the per-bucket overlap report is there."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["rccl_ranks"]["world_size"] == 2 and line["rccl_ranks"]["allreduce_of_ones"] == 2
    assert line["comm_overlap"]["buckets"], line["comm_overlap"]


def test_bench_reports_comm_overlap_on_a_one_rank_group():
    """bench.py's data-parallel reporting on ONE GPU (1-rank RCCL group, CRDR_FORCE_DIST=1): the line carries `rccl_ranks` and the
    per-bucket overlap trace of the staged generator step -- three generator buckets in backward order + the active sub-discriminator's
    -- with every all-reduce hidden behind compute (the events that a SCALE run will show at N = 2, 4, 8)."""
    import json
    env = dict(os.environ, CRDR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29688")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary",
                        "--profile-steps", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), out_lines   # exactly the JSON line: RCCL's version banner goes to stderr
    line = json.loads(out_lines[0])
    assert line["rccl_ranks"]["backend"] == "nccl" and line["rccl_ranks"]["allreduce_of_ones"] == 1
    assert line["ms_per_step_median"] and line["value_at_median"]
    names = [b["bucket"] for b in line["comm_overlap"]["buckets"]]
    assert names == ["g.decoder", "g.context_model", "g.rest", "d"], names
    mb = {b["bucket"]: b["MB"] for b in line["comm_overlap"]["buckets"]}
    assert abs(mb["g.decoder"] + mb["g.context_model"] + mb["g.rest"] - 510.8) < 1.0 and abs(mb["d"] - 18.8) < 0.2, mb
    assert all(b["allreduce_ms"] > 0 and b["exposed_ms"] >= 0 for b in line["comm_overlap"]["buckets"])
