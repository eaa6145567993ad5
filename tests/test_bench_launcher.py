"""bench.py started without a launcher (`python bench.py --gpus N`, the way the driver calls it) must start its N ranks itself,
as a CHILD torch.distributed.run, before anything touches the GPU, and relay rank 0's JSON line."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launcher_argv_is_the_drivers_torchrun_line():
    argv = bench.launcher_argv(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "29511"
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_self_launch_relays_rank0_line(monkeypatch, capsys):
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        out = "noise from a rank\n" + json.dumps({"metric": "stage-3 training img/s at 256x256", "value": 1.0, "n_gpus": 2}) + "\n"
        return types.SimpleNamespace(stdout=out, returncode=0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.self_launch(2, ["--gpus", "2"])
    assert rc == 0
    assert "--nproc-per-node=2" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_main_self_launches_before_importing_torch(monkeypatch):
    """With --gpus 2 and no WORLD_SIZE, main() must go to self_launch without initialising anything."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    called = {}
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: called.setdefault("n", (n, list(argv))) and 0)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called["n"] == (2, ["--gpus", "2", "--steps", "1"])


def test_world_size_mismatch_is_a_clear_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        bench.main()
        assert False, "expected SystemExit"
    except SystemExit as e:
        assert "WORLD_SIZE=1" in str(e.code)


def test_train_data_generator_advances_the_sampler_epoch():
    """a DistributedSampler(shuffle=True) replays one permutation unless set_epoch() is called per pass."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler
    from crdr_amd.trainer.base_trainer import BaseTrainer
    ds = TensorDataset(torch.arange(16))
    sampler = DistributedSampler(ds, num_replicas=2, rank=0, shuffle=True, seed=3, drop_last=True)
    dl = DataLoader(ds, batch_size=4, sampler=sampler, drop_last=True)   # 2 batches per epoch for this rank
    gen = BaseTrainer.train_data_generator(None, dl, 0, 6)
    epochs = [[], [], []]
    for i, (b,) in gen:
        epochs[(i - 1) // 2] += b.tolist()
    assert sorted(epochs[0]) != sorted(epochs[1]) or epochs[0] != epochs[1]
    assert epochs[0] != epochs[1] and epochs[1] != epochs[2]
    # a run resumed at iteration 4 continues with epoch 2's order
    resumed = [b.tolist() for _, (b,) in BaseTrainer.train_data_generator(None, dl, 4, 6)]
    assert sum(resumed, []) == epochs[2]
