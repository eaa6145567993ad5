"""Child process of tests/test_gpu_dp.py: runs the REAL stage-3 trainer for a few iterations on seeded weights and inputs
and dumps every logged scalar plus the final parameters.  With CRDR_FORCE_DIST=1 in the environment the trainer takes
its data-parallel path -- a 1-rank RCCL process group, flat-buffer all-reduces on the communication stream, the MAX
-reduced skip flag, shared (q, beta) draws -- which must not change a single bit with respect to the plain run.

    python -m tests.dp_step_worker OUT.pt [--graphs] [--iters N]

Under a launcher (`python -m torch.distributed.run --nproc-per-node W -m tests.dp_step_worker OUT.pt --shard ...`, BASELINE config #4
on W GPUs) every rank takes cuda:LOCAL_RANK and its own slice of a seeded GLOBAL batch (images and explicit per-sample noise), and
writes OUT.pt.rank<r>: the test compares the ranks with each other and with the one-GPU run of the whole batch."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("CRDR_ALLOW_RANDOM_LPIPS", "1")


def main():
    """One variant (`OUT.pt [options]`) or several in ONE process (`--batch SPEC.json`: a list of {"out": path, "args": [options]}): the
    GPU suite's child-process tests used to start 18 interpreters of ~10 s each (import, library load, trainer build) for variants that
    differ in a flag; variants of one kind (all plain, or all on the 1-rank process group) now share a process, each with a fresh
    trainer, freshly seeded generators and its own dump."""
    if len(sys.argv) >= 3 and sys.argv[1] == "--batch":
        import gc
        import json
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 16))   # (the trainer builds -- host-side initialisation -- are this process's CPU work: tests/conftest.py)
        with open(sys.argv[2]) as f:
            specs = json.load(f)
        for spec in specs:
            run_one([spec["out"]] + list(spec["args"]), last=spec is specs[-1])
            gc.collect()
            torch.cuda.empty_cache()
        return
    run_one(sys.argv[1:], last=True)


_dist_ready = []


def run_one(argv, last=True):
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--graphs", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--draw-conditions", action="store_true", help="let the trainer draw (q, beta) itself")
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--no-buckets", action="store_true", help="data-parallel path with ONE gradient bucket after an unstaged backward")
    ap.add_argument("--target-rate", type=float, default=None, help="override the rate loss' target(s): the lambda_A / lambda_B switch goes live")
    ap.add_argument("--shard", action="store_true", help="global batch of --global-bs seeded images + explicit noise; this rank trains on its slice")
    ap.add_argument("--global-bs", type=int, default=4)
    ap.add_argument("--fixed-q", type=int, default=None, help="one rate index for every iteration")
    ap.add_argument("--forced-algo", type=int, default=0, help="ops.FORCED_CONV_ALGO: one conv plan whatever the batch size (runs at different "
                    "per-process batch sizes then share the fp32 summation order)")
    ap.add_argument("--report-qbpp", action="store_true", help="no training: per-image quantised bpp of the global batch at --fixed-q")
    ap.add_argument("--straddle-target", action="store_true", help="target rate between the two half-batch means of the quantised bpp of the "
                    "global batch (computed here, before training, from the seeded parameters): the lambda_A / lambda_B switch is straddled")
    a = ap.parse_args(argv)
    import torch
    from crdr_amd.trainer import build_trainer
    from crdr_amd.trainer import dist as D
    from tests.golden.seeded_weights import seeded_input
    from tests.test_gpu_step import _opt, _seed_params
    from crdr_amd.hip import ops
    if not _dist_ready:
        _dist_ready.append(D.init_from_env())
    local = _dist_ready[0]
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    ops.FORCED_CONV_ALGO = a.forced_algo
    torch.manual_seed(0)  # the factorised prior's noise comes from torch's (graph-safe) CUDA generator
    ws, rk = (D.world_size(), D.rank()) if a.shard else (1, 0)
    assert a.global_bs % ws == 0
    per = a.global_bs // ws if a.shard else 2
    opt = _opt(a.stage, bs=per)
    opt["device"] = device
    opt["hip_graphs"] = a.graphs
    opt["hip_graph_warmup"] = 1
    opt["dp_buckets"] = not a.no_buckets
    tr = build_trainer(opt)
    _seed_params(tr.comp_model, "")
    if a.stage == 3:
        _seed_params(tr.discriminator, "")
    if tr.perceptual_loss is not None:
        _seed_params(tr.perceptual_loss.lpips, "lpips.")
    if a.target_rate is not None:
        rl = tr.rate_loss
        rl.target_rate = [a.target_rate] * len(rl.target_rate) if isinstance(rl.target_rate, list) else a.target_rate
    tr.comp_model.context_model.seed_noise(1234)
    tr.loss_huge_threshold = float("inf")
    noise = None
    if a.shard:
        sl = slice(rk * per, (rk + 1) * per)
        x = seeded_input("dp.image", (a.global_bs, 3, 64, 64))[sl].to(device)
        noise = {"y": seeded_input("dp.noise.y", (a.global_bs, 320, 4, 4), 0.5)[sl].to(device),
                 "z": seeded_input("dp.noise.z", (a.global_bs, 192, 1, 1), 0.5)[sl].to(device)}
    else:
        x = seeded_input("image", (2, 3, 64, 64)).to(device)
    target_used = a.target_rate
    if a.straddle_target:   # (needs --shard --fixed-q: the whole global batch through the seeded model, no training)
        gx = seeded_input("dp.image", (a.global_bs, 3, 64, 64)).to(device)
        gn = {"y": seeded_input("dp.noise.y", (a.global_bs, 320, 4, 4), 0.5).to(device),
              "z": seeded_input("dp.noise.z", (a.global_bs, 192, 1, 1), 0.5).to(device)}
        with torch.no_grad():
            qb = tr.comp_model.run_model(gx, rate_ind=float(a.fixed_q), beta=2.56, is_train=True, noise=gn)["qbpp"].detach().double().cpu()
        lo, hi = sorted([float(qb[: a.global_bs // 2].mean()), float(qb[a.global_bs // 2:].mean())])
        target_used = lo + 0.25 * (hi - lo)
        rl = tr.rate_loss
        rl.target_rate = [target_used] * len(rl.target_rate) if isinstance(rl.target_rate, list) else target_used
        tr.comp_model.context_model.seed_noise(1234)
        torch.manual_seed(0)
    if a.report_qbpp:
        with torch.no_grad():
            out = tr.comp_model.run_model(x, rate_ind=float(a.fixed_q), beta=2.56, is_train=True, noise=noise)
        torch.save({"qbpp": out["qbpp"].detach().cpu()}, a.out)
        return
    grads = {}
    g_step = tr.g_optimizer.step

    def spy_g(*args, **kw):   # the first (eager) updates: gradients as the optimisers see them, i.e. after their all-reduce
        if "G" not in grads:
            grads["G"] = torch.cat([b.reshape(-1) for b in tr.g_optimizer.flat_grads()]).detach().cpu()
        return g_step(*args, **kw)
    tr.g_optimizer.step = spy_g
    if a.stage == 3:
        d_step = tr.d_optimizer.step

        def spy_d(*args, **kw):
            if "D" not in grads:
                grads["D"] = torch.cat([b.reshape(-1) for b in tr.d_optimizer.flat_grads(partitions=kw.get("partitions"))]).detach().cpu()
            return d_step(*args, **kw)
        tr.d_optimizer.step = spy_d
    logs, qs = [], []
    for it in range(1, a.iters + 1):
        data = {"real_images": x}
        if noise is not None:
            data["noise"] = noise
        if not a.draw_conditions and a.stage == 3:
            qs.append(a.fixed_q if a.fixed_q is not None else it % 2 + 1)
            data.update(rate_ind=qs[-1], beta=2.56 + 0.01 * it)
        logs.append(tr.optimize_parameters(it, data))
    torch.cuda.synchronize()
    state = {"logs": logs, "dist": D.is_dist(), "world": D.world_size(), "rank": D.rank(), "grads": grads,
             "g_layout": [(n, p.numel()) for n, p in _flat_order(tr)],
             "G": {k: p.detach().cpu() for k, p in tr.comp_model.named_parameters()},
             "D": {k: p.detach().cpu() for k, p in tr.discriminator.named_parameters()} if a.stage == 3 else {},
             "staged": bool(tr._staged()),
             "graphs": len(tr.graphs), "target": target_used,
             "probe_qbpp": qb if a.straddle_target else None}
    torch.save(state, a.out + (f".rank{D.rank()}" if a.shard and D.is_dist() else ""))
    tr.g_optimizer.step = g_step
    if last and torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _flat_order(tr):
    """(name, parameter) in the order of the generator optimiser's flat gradient buffers"""
    names = {id(p): n for n, p in tr.comp_model.named_parameters()}
    return [(names[id(p)], p) for g in tr.g_optimizer.param_groups if g["grad"] is not None for p in g["params"]]


if __name__ == "__main__":
    main()
