"""Data-parallel logic on CPU with the gloo backend (world_size 2 and 4): flat-buffer averaging, rank-consistent skip
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
        mean_rank1 = (world + 1) / 2.0   # mean of rank + 1 over the ranks
        bufs = [torch.full((5,), float(rank + 1)), torch.arange(3.0) * (rank + 1)]
        D.all_reduce_mean_(bufs)
        assert torch.allclose(bufs[0], torch.full((5,), mean_rank1)) and torch.allclose(bufs[1], torch.arange(3.0) * mean_rank1)
        # 1b) the asynchronous form the stage-3 trainer uses (gradient buffers mean-reduced, skip flag max-reduced)
        gb, flag = torch.full((7,), float(2 * rank)), torch.tensor([float(rank)])
        sync = D.AsyncGradSync([gb], [flag])
        sync.wait()
        assert torch.allclose(gb, torch.full((7,), float(world - 1))) and float(flag) == float(world - 1)
        # 1c) the mean is exact for rank counts that are not 2: a value every rank holds comes back bit for bit (world a power of two: the
        # division is exact; the sum of equal addends is exact below 2^24 x ulp)
        same = torch.tensor([0.1, 1.0 / 3.0, 1e-7, 123456.789])
        keep = same.clone()
        D.all_reduce_mean_([same])
        assert torch.equal(same, keep), (same, keep)
        # 2) skip decision is an OR over ranks; scalar mean
        assert D.any_rank_true(rank == 1, torch.device("cpu")) is True
        assert D.any_rank_true(rank == world - 1, torch.device("cpu")) is True
        assert D.any_rank_true(False, torch.device("cpu")) is False
        assert abs(float(D.all_reduce_scalars_mean(torch.tensor(float(rank)))) - (world - 1) / 2.0) < 1e-6
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
        per = 4 // world
        loss_fn(mine, x[rank * per:(rank + 1) * per]).backward()
        flat = torch.cat([mine[k].grad.reshape(-1) for k in sorted(mine)])
        D.all_reduce_mean_([flat])
        ref_flat = torch.cat([full[k].grad.reshape(-1) for k in sorted(full)])
        assert torch.allclose(flat, ref_flat, rtol=1e-5, atol=1e-6)
        # 5) the trainer's staged generator step (forward | global mean qbpp | backward in three pieces, one asynchronous
        # gradient bucket each) driven through the REAL trainer methods on a toy CPU model with the product's cut mechanism
        _staged_step_check(rank, world, D)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _staged_step_check(rank, world, D):
    import types
    import torch.nn as nn
    from crdr_amd.losses.rate_loss import HificRateLoss
    from crdr_amd.models.comp_model.hyperprior_charm_model import HyperpriorCharmModel
    from crdr_amd.trainer.base_trainer import BaseTrainer
    from crdr_amd.trainer.rate_distortion_trainer import RateDistortionTrainer

    class Toy(nn.Module):
        """encoder -> hyper pair -> 'context model' (bits + y_hat) -> decoder, cut like HyperpriorCharmModel.forward"""
        backward_cuts = None
        _cut = HyperpriorCharmModel._cut

        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.encoder, self.hyperencoder = nn.Conv2d(3, 4, 3, padding=1), nn.Conv2d(4, 2, 3, padding=1)
            self.hyperdecoder, self.context_model = nn.Conv2d(2, 4, 3, padding=1), nn.Conv2d(8, 4, 3, padding=1)
            self.decoder = nn.Conv2d(4, 3, 3, padding=1)

        def run_model(self, real_images):
            y = self.encoder(real_images)
            z = self.hyperencoder(y)
            hyper = self.hyperdecoder(z)
            bits_z = z.square().sum((1, 2, 3))
            y_in, h_in = self._cut("y", y, True), self._cut("hyper_out", hyper, True)
            mu = self.context_model(torch.cat([y_in, h_in], 1))
            bits_y = (y_in - mu).square().sum((1, 2, 3))
            y_hat = y_in + 0.1 * mu
            fake = self.decoder(self._cut("y_hat", y_hat, True))
            npix = real_images.shape[2] * real_images.shape[3]
            return dict(real_images=real_images, fake_images=fake, bpp=(bits_y + bits_z) / npix, qbpp=((bits_y + bits_z) / npix).detach(),
                        bits_y=bits_y, bits_z=bits_z, num_pixel=npix)

    def make(x):
        tr = RateDistortionTrainer.__new__(RateDistortionTrainer)
        tr.comp_model = Toy()
        named = dict(sorted(tr.comp_model.named_parameters()))
        flat = torch.zeros(sum(p.numel() for p in named.values()))
        off = 0
        for p in named.values():
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        tr.g_optimizer = types.SimpleNamespace(param_groups=[{"params": list(named.values()), "grad": flat}], zero_grad=lambda: flat.zero_())
        tr.aux_optimizer, tr.perceptual_loss, tr._pieces = None, None, None
        tr.distortion_loss = lambda real, fake, **kw: (real - fake).square().mean()
        tr.rate_loss = HificRateLoss(lambda_A=2.0, lambda_B=0.5, target_rate=target)
        tr.opt = {"dp_buckets": True}
        tr.loss_huge_threshold = 1e4
        tr._flush_wgrads = lambda site: None
        return tr, flat

    torch.manual_seed(11)
    x = torch.randn(4, 3, 8, 8)
    x[2:] *= 3.0   # the upper half of the batch (rank 1 of 2, ranks 2 and 3 of 4) has the larger rate: the local means straddle the target
    probe, _ = None, None
    with torch.no_grad():
        out = Toy().run_model(x)
    q_all = out["qbpp"]
    lo, hi, mid = float(q_all[:2].mean()), float(q_all[2:].mean()), float(q_all.mean())
    assert lo < hi
    for target in (0.5 * (lo + mid), 0.5 * (mid + hi)):   # global mean above / below the target; one rank on each side locally
        tr, flat = make(x)
        assert tr._staged()
        pieces = tr._piece_buffers()
        assert [sum(b.numel() for b in p) for p in pieces] == [sum(q.numel() for q in m.parameters()) for m in
                                                               (tr.comp_model.decoder, tr.comp_model.context_model)] + \
            [sum(q.numel() for n, q in tr.comp_model.named_parameters() if n.split(".")[0] not in ("decoder", "context_model"))]
        per = 4 // world
        ctx, syncs = tr._run_generator_staged(lambda name, fn: fn(), x[rank * per:(rank + 1) * per], {}, None, 1)
        for sy in syncs:
            sy.wait()
        # reference: one unstaged backward over the whole batch with the switch on the global mean
        ref, ref_flat = make(x)
        ref._staged = lambda: False
        f = ref._g_forward(x, {}, None, 1)
        rate = ref.rate_loss(f["bpp"], **f["other"], current_iter=1)
        (f["nonrate"] + rate).backward()
        want_lambda = 2.0 if mid > target else 0.5
        assert abs(float(rate) - want_lambda * mid) < 1e-5 * max(1.0, mid)
        assert torch.allclose(flat, ref_flat, rtol=1e-4, atol=1e-6), (target, float((flat - ref_flat).abs().max()))
        assert float(ctx["bad"]) == 0.0 and abs(float(ctx["losses"]["rate"]) - want_lambda * float(q_all[rank * per:(rank + 1) * per].mean())) < 1e-4 * hi


def _run_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_dp_two_ranks_gloo():
    _run_world(2)


def test_dp_four_ranks_gloo():
    """the same checks on four ranks: shards of one image each, bucket partition and AVG for a rank count that is not 2, the rate switch
    straddled by ranks 0 / 1 (below) and 2 / 3 (above the target)"""
    _run_world(4)
