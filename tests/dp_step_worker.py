"""Child process of tests/test_gpu_dp.py: runs the REAL stage-3 trainer for a few iterations on seeded weights and inputs
and dumps every logged scalar plus the final parameters.  With CRDR_FORCE_DIST=1 in the environment the trainer takes
its data-parallel path -- a 1-rank RCCL process group, flat-buffer all-reduces on the communication stream, the MAX
-reduced skip flag, shared (q, beta) draws -- which must not change a single bit with respect to the plain run.

    python -m tests.dp_step_worker OUT.pt [--graphs] [--iters N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("CRDR_ALLOW_RANDOM_LPIPS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--graphs", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--draw-conditions", action="store_true", help="let the trainer draw (q, beta) itself")
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--no-buckets", action="store_true", help="data-parallel path with ONE gradient bucket after an unstaged backward")
    ap.add_argument("--target-rate", type=float, default=None, help="override the rate loss' target(s): the lambda_A / lambda_B switch goes live")
    a = ap.parse_args()
    import torch
    from crdr_amd.trainer import build_trainer
    from crdr_amd.trainer import dist as D
    from tests.golden.seeded_weights import seeded_input
    from tests.test_gpu_step import _opt, _seed_params
    local = D.init_from_env()
    torch.cuda.set_device(local)
    torch.manual_seed(0)  # the factorised prior's noise comes from torch's (graph-safe) CUDA generator
    opt = _opt(a.stage)
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
    x = seeded_input("image", (2, 3, 64, 64)).to("cuda:0")
    logs = []
    for it in range(1, a.iters + 1):
        data = {"real_images": x}
        if not a.draw_conditions and a.stage == 3:
            data.update(rate_ind=it % 2 + 1, beta=2.56 + 0.01 * it)
        logs.append(tr.optimize_parameters(it, data))
    torch.cuda.synchronize()
    state = {"logs": logs, "dist": D.is_dist(), "world": D.world_size(),
             "G": {k: p.detach().cpu() for k, p in tr.comp_model.named_parameters()},
             "D": {k: p.detach().cpu() for k, p in tr.discriminator.named_parameters()} if a.stage == 3 else {},
             "staged": bool(tr._staged()),
             "graphs": len(tr.graphs)}
    torch.save(state, a.out)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
