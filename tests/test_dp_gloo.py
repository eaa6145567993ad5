"""Data-parallel logic on CPU with the gloo backend (world_size 2): flat-buffer averaging, rank-consistent skip
decisions, identical per-iteration (q, beta) draws on every rank, and the DP identity itself -- averaging the
per-rank gradients of the step's loss over shards equals the single-process gradient over the whole batch
(computed with the CPU oracle, so no GPU is needed)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from crdr_amd.trainer import dist as D
    D.init_from_env(backend="gloo")
    try:
        assert D.is_dist() and D.world_size() == world and D.rank() == rank
        # 1) flat-buffer mean
        bufs = [torch.full((5,), float(rank + 1)), torch.arange(3.0) * (rank + 1)]
        D.all_reduce_mean_(bufs)
        assert torch.allclose(bufs[0], torch.full((5,), 1.5)) and torch.allclose(bufs[1], torch.arange(3.0) * 1.5)
        # 1b) the asynchronous form the stage-3 trainer uses (gradient buffers mean-reduced, skip flag max-reduced)
        gb, flag = torch.full((7,), float(2 * rank)), torch.tensor([float(rank)])
        sync = D.AsyncGradSync([gb], [flag])
        sync.wait()
        assert torch.allclose(gb, torch.full((7,), 1.0)) and float(flag) == 1.0
        # 2) skip decision is an OR over ranks; scalar mean
        assert D.any_rank_true(rank == 1, torch.device("cpu")) is True
        assert D.any_rank_true(False, torch.device("cpu")) is False
        assert abs(float(D.all_reduce_scalars_mean(torch.tensor(float(rank)))) - 0.5) < 1e-6
        # 3) identical condition draws (same seeded generators as the stage-3 trainer)
        import numpy as np
        g = torch.Generator().manual_seed(0)
        rng = np.random.default_rng(0)
        draws = torch.tensor([[float(torch.randint(5, (1,), generator=g)), float(rng.integers(0, 101))] for _ in range(16)])
        ref = draws.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(draws, ref)
        # 4) DP identity with a small conv net under the oracle's loss pieces: mean over shards == full batch
        from oracle import crdr_oracle as O
        torch.manual_seed(0)
        w = {"c.weight": torch.randn(8, 3, 3, 3) * 0.2, "c.bias": torch.randn(8) * 0.1}
        x = torch.randn(4, 3, 16, 16)

        def loss_fn(sd, xb):
            y = O.conv(sd, "c", xb, pad=1)
            return O.mse_loss(xb[:, :3], y[:, :3]) + O.gan_loss(y[:, 3:4], True, True, 1.0)
        full = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        loss_fn(full, x).backward()
        mine = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        loss_fn(mine, x[rank * 2:(rank + 1) * 2]).backward()
        flat = torch.cat([mine[k].grad.reshape(-1) for k in sorted(mine)])
        D.all_reduce_mean_([flat])
        ref_flat = torch.cat([full[k].grad.reshape(-1) for k in sorted(full)])
        assert torch.allclose(flat, ref_flat, rtol=1e-5, atol=1e-6)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_dp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
