"""HIP-graph execution of the step equals eager execution: two trainers with identical seeded weights and identical
(q, beta, noise, batch) run five iterations, one launching every kernel from Python, the other replaying captured
graphs (after its two eager warm-up iterations).  Same kernels, same order, deterministic reductions => the logged
losses and every parameter must agree bit for bit; the skip gate must hold parameters still."""
import pytest
import torch

from tests.golden.seeded_weights import seeded_input
from tests.test_gpu_model import dev
from tests.test_gpu_step import _opt, _seed_params

pytestmark = pytest.mark.gpu


def _trainer(stage, graphs):
    from crdr_amd.trainer import build_trainer
    opt = _opt(stage)
    opt["hip_graphs"] = graphs
    tr = build_trainer(opt)
    _seed_params(tr.comp_model, "")
    if stage == 3:
        _seed_params(tr.discriminator, "")
    _seed_params(tr.perceptual_loss.lpips, "lpips.")
    tr.loss_huge_threshold = float("inf")
    return tr


@pytest.mark.parametrize("stage", [3, 1])
def test_graph_replay_equals_eager(stage):
    x = seeded_input("image", (2, 3, 64, 64)).to(dev())
    noise = {"y": seeded_input("noise.y", (2, 320, 4, 4), 0.5).to(dev()), "z": seeded_input("noise.z", (2, 192, 1, 1), 0.5).to(dev())}
    logs = {}
    params = {}
    for mode in (False, True):
        tr = _trainer(stage, mode)
        out = []
        for it in range(1, 6):
            data = {"real_images": x, "noise": noise}
            if stage == 3:
                data.update(rate_ind=1, beta=2.56 + 0.01 * it)  # beta changes every step: must flow through device buffers
            out.append(tr.optimize_parameters(it, data))
        if mode:
            assert len(tr.graphs) == (4 if stage == 3 else 2), "segments were not captured"  # g, dfb, u, d
        logs[mode] = out
        params[mode] = {k: p.detach().clone() for k, p in tr.comp_model.named_parameters()}
        if stage == 3:
            params[mode].update({"D." + k: p.detach().clone() for k, p in tr.discriminator.named_parameters()})
        del tr
    for a, b in zip(logs[False], logs[True]):
        assert a is not None and b is not None
        assert a.keys() == b.keys()
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])
    for k in params[False]:
        assert torch.equal(params[False][k], params[True][k]), k


def test_skip_gate_inside_graph():
    tr = _trainer(3, True)
    tr.loss_huge_threshold = 10000.0  # seeded weights give a loss far above it -> every iteration must be skipped
    x = seeded_input("image", (2, 3, 64, 64)).to(dev())
    before = {k: p.detach().clone() for k, p in tr.comp_model.named_parameters()}
    for it in range(1, 5):
        assert tr.optimize_parameters(it, {"real_images": x, "rate_ind": 2, "beta": 1.0}) is None
    for k, p in tr.comp_model.named_parameters():
        assert torch.equal(before[k], p.detach()), k
    for k, p in tr.discriminator.named_parameters():
        assert torch.isfinite(p).all()
    tr.g_optimizer.host_step_counts()
    assert tr.g_optimizer.param_groups[0]["step"] == 0


def test_filter_caches_created_after_capture_follow_the_replayed_updates():
    """Persistent F(4x4) filter caches are valid by the version counters of their weight packs, and those counters used to advance only when
    Python ran PackTable.refill -- which a captured optimiser segment does not.  A cache created AFTER the capture (an eager evaluation pass at
    another image size) was filled once, stamped, and then trusted forever while the replays went on updating the packs on the device.  Now
    SegmentGraphs.run calls the refill's host-side bookkeeping after every replay (ops.on_replay / PackTable._replayed): the packs' versions
    advance, a cache the replayed rebuild launch did not cover falls behind and re-transforms, and it joins the device table for the next
    replay.  Checked against a forced re-transform, bit for bit."""
    from crdr_amd.hip import ops
    ops.PREFER_WINOGRAD = 4
    try:
        tr = _trainer(1, True)
        x = seeded_input("image", (2, 3, 64, 64)).to(dev())
        xe = seeded_input("eval image", (1, 3, 128, 128)).to(dev())
        noise = {"y": seeded_input("noise.y", (2, 320, 4, 4), 0.5).to(dev()), "z": seeded_input("noise.z", (2, 192, 1, 1), 0.5).to(dev())}
        for it in range(1, 5):
            assert tr.optimize_parameters(it, {"real_images": x, "noise": noise}) is not None
        assert len(tr.graphs) == 2
        n0 = len(ops._filter_cache)
        assert n0 > 0, "the warm-up created no F(4x4) filter cache: the test would be vacuous"
        def evaluate():
            with torch.no_grad():
                return tr.comp_model.run_model(xe, is_train=False)["fake_images"].clone()
        out1 = evaluate()
        assert len(ops._filter_cache) > n0, "the evaluation shapes created no new cache"
        st0 = dict(ops.FILTER_SCOPE_STATS)
        for it in range(5, 8):   # replays only: the packs change on the device
            assert tr.optimize_parameters(it, {"real_images": x, "noise": noise}) is not None
        out2 = evaluate()
        ops.filter_scope_invalidate()    # every cache dropped: the next pass transforms from the current packs
        out3 = evaluate()
        assert not torch.equal(out1, out2), "three optimiser steps must have moved the reconstruction"
        assert torch.equal(out2, out3), "an evaluation pass after graph replays ran on filters transformed from OLD weights"
        assert ops.FILTER_SCOPE_STATS["batched"] > st0["batched"] or ops.FILTER_SCOPE_STATS["filled"] > st0["filled"]
    finally:
        ops.PREFER_WINOGRAD = False
        ops.filter_scope_invalidate()
