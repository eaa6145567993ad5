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
    a = ap.parse_args()
    import torch
    from crdr_amd.trainer import build_trainer
    from crdr_amd.trainer import dist as D
    from tests.golden.seeded_weights import seeded_input
    from tests.test_gpu_step import _opt, _seed_params
    local = D.init_from_env()
    torch.cuda.set_device(local)
    torch.manual_seed(0)  # the factorised prior's noise comes from torch's (graph-safe) CUDA generator
    opt = _opt(3)
    opt["hip_graphs"] = a.graphs
    opt["hip_graph_warmup"] = 1
    tr = build_trainer(opt)
    _seed_params(tr.comp_model, "")
    _seed_params(tr.discriminator, "")
    _seed_params(tr.perceptual_loss.lpips, "lpips.")
    tr.comp_model.context_model.seed_noise(1234)
    tr.loss_huge_threshold = float("inf")
    x = seeded_input("image", (2, 3, 64, 64)).to("cuda:0")
    logs = []
    for it in range(1, a.iters + 1):
        data = {"real_images": x}
        if not a.draw_conditions:
            data.update(rate_ind=it % 2 + 1, beta=2.56 + 0.01 * it)
        logs.append(tr.optimize_parameters(it, data))
    torch.cuda.synchronize()
    state = {"logs": logs, "dist": D.is_dist(), "world": D.world_size(),
             "G": {k: p.detach().cpu() for k, p in tr.comp_model.named_parameters()},
             "D": {k: p.detach().cpu() for k, p in tr.discriminator.named_parameters()},
             "graphs": len(tr.graphs)}
    torch.save(state, a.out)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
