"""End-to-end drop-in check of the two entry points on the GPU (scripts/train.py:16-27, scripts/compress.py:35-47):
train stage 3 for a few iterations on a tiny PNG folder (host DataLoader and the device-pool pipeline), with HIP graphs,
logging, validation and checkpoints; resume from the checkpoint; then compress / decompress an image folder with the
trained weights and check the container accounting."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _png_dir(path, n, h, w, seed):
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n):
        base = rng.integers(0, 256, size=(h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        img = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w]  # blocky, compressible
        Image.fromarray(img).save(os.path.join(path, f"im{i:02d}.png"))


def _run(args, cwd, timeout=900):
    env = dict(os.environ, CRDR_AUTOTUNE="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


@pytest.mark.parametrize("device_pool", [False, True])
def test_train_resume_compress(tmp_path, device_pool):
    train_dir, eval_dir = str(tmp_path / "train" / "0"), str(tmp_path / "kodak")
    _png_dir(train_dir, 6, 96, 120, 1)
    _png_dir(eval_dir, 2, 64, 128, 2)
    os.makedirs(tmp_path / "checkpoint")  # like the reference, ckpt_root must exist (options.py:174-177)
    cfg = tmp_path / "tiny_stage3.yaml"
    cfg.write_text(f"""_base_: [{os.path.relpath(os.path.join(ROOT, 'config', 'crdr_stage_3.yaml'), str(tmp_path))}]
pretrained_weight_path: null
ckpt_root: {tmp_path}/checkpoint
hip_graphs: true
keep_training_state: true
keep_discriminator: true
dataset:
  batch_size: 2
  train_dataset:
    root_dir: {tmp_path}/train
    name: openimage
    type: ImageDataset
    image_size: 64
    subset_list: [0]
    device_pool: {'true' if device_pool else 'false'}
  eval_dataset:
    root_dir: {eval_dir}
    name: Kodak
    type: ImageDataset
""")
    train = os.path.join(ROOT, "scripts", "train.py")
    _run([train, str(cfg), "-d", "cuda:0", "-b", "2", "-ti", "6", "-s", "3", "-l", "2", "-e", "3", "-nw", "0"], cwd=str(tmp_path))
    model_dir = tmp_path / "checkpoint" / "tiny_stage3" / "model"
    names = sorted(os.listdir(model_dir))
    assert "comp_model_iter6.pth.tar" in names and any(n.startswith("training_state_iter6") for n in names), names
    # resume: two more iterations from the iteration-6 checkpoint
    _run([train, str(cfg), "-d", "cuda:0", "-b", "2", "-si", "6", "-ti", "8", "-s", "2", "-l", "1", "-e", "100", "-nw", "0"], cwd=str(tmp_path))
    assert "comp_model_iter8.pth.tar" in os.listdir(model_dir)
    # compress / decompress with the trained generator
    out_dir = tmp_path / "out"
    _run([os.path.join(ROOT, "scripts", "compress.py"), "--config_path", os.path.join(ROOT, "config", "crdr.yaml"), "--model_path",
          str(model_dir / "comp_model_iter8.pth.tar"), "--img_dir", eval_dir, "--save_dir", str(out_dir), "-q", "1.5", "-b", "2.56",
          "--decompress", "-d", "cuda:0"], cwd=ROOT)
    files = sorted(os.listdir(out_dir))
    assert "im00.bin" in files and "im00.png" in files and "_avg_bitrate.json" in files, files
    avg = json.load(open(out_dir / "_avg_bitrate.json"))
    assert 0 < list(avg.values())[0] < 24
    from PIL import Image
    assert Image.open(out_dir / "im00.png").size == (128, 64)


def test_three_stage_recipe(tmp_path):
    """The reference's training recipe end to end at toy scale: stage 1 (single rate) -> stage 2 (multi-rate InterpCA,
    initialised from stage 1 through `pretrained_weight_path`, non-strict like the reference) -> stage 3 (GAN, initialised
    from stage 2); every hand-over goes through the checkpoint files scripts/train.py writes."""
    train_dir, eval_dir = str(tmp_path / "train" / "0"), str(tmp_path / "kodak")
    _png_dir(train_dir, 4, 80, 96, 3)
    _png_dir(eval_dir, 1, 64, 64, 4)
    os.makedirs(tmp_path / "checkpoint")
    train = os.path.join(ROOT, "scripts", "train.py")
    prev = None
    for stage in (1, 2, 3):
        cfg = tmp_path / f"toy_stage{stage}.yaml"
        cfg.write_text(f"""_base_: [{os.path.relpath(os.path.join(ROOT, 'config', f'crdr_stage_{stage}.yaml'), str(tmp_path))}]
pretrained_weight_path: {prev if prev else 'null'}
ckpt_root: {tmp_path}/checkpoint
hip_graphs: true
dataset:
  batch_size: 2
  train_dataset:
    root_dir: {tmp_path}/train
    name: openimage
    type: ImageDataset
    image_size: 64
    subset_list: [0]
  eval_dataset:
    root_dir: {eval_dir}
    name: Kodak
    type: ImageDataset
""")
        _run([train, str(cfg), "-d", "cuda:0", "-b", "2", "-ti", "4", "-s", "4", "-l", "2", "-e", "4", "-nw", "0"], cwd=str(tmp_path))
        prev = str(tmp_path / "checkpoint" / f"toy_stage{stage}" / "model" / "comp_model_iter4.pth.tar")
        assert os.path.exists(prev), os.listdir(os.path.dirname(prev))


def test_compress_demo_images(tmp_path):
    """BASELINE config #1 (plumbing): scripts/compress.py over ./demo_images (three Kodak images, public data) at q = 0,
    beta = 3.84 -- here on the GPU and with random-init weights (the reference's checkpoint is a Google-Drive download), so
    the check is the container: one .bin + one 768x512 .png per image, per-image accounting, average bpp file."""
    import pandas as pd
    from PIL import Image
    out_dir = tmp_path / "out"
    _run([os.path.join(ROOT, "scripts", "compress.py"), "--config_path", os.path.join(ROOT, "config", "crdr.yaml"), "--img_dir",
          os.path.join(ROOT, "demo_images"), "--save_dir", str(out_dir), "-q", "0.0", "-b", "3.84", "--decompress", "-d", "cuda:0"],
         cwd=ROOT)
    names = ["kodim03", "kodim15", "kodim23"]
    for n in names:
        assert os.path.exists(out_dir / f"{n}.bin")
        assert Image.open(out_dir / f"{n}.png").size == (768, 512)
    df = pd.read_csv(out_dir / "_bitrates.csv")
    assert sorted(df["img_name"]) == [n + ".png" for n in names]
    for _, r in df.iterrows():
        nbytes = os.path.getsize(out_dir / r["img_name"].replace(".png", ".bin"))
        assert r["real_bit"] == 8 * nbytes == r["header_bit"] + r["z_bit"] + r["y_bit"] + 96  # 3 x u32 length prefixes
        assert abs(r["real_bpp"] - 8 * nbytes / (768 * 512)) < 1e-12 and r["header_bit"] == 48
    avg = json.load(open(out_dir / "_avg_bitrate.json"))
    assert abs(list(avg.values())[0] - df["real_bpp"].mean()) < 1e-9


def test_stage3_with_hific_discriminator(tmp_path):
    """Registry / YAML surface: swapping the discriminator type in the config is all it takes to train stage 3 against the
    spectrally normalised HiFiC discriminator (graphs, deferred reductions and the power iteration included)."""
    train_dir, eval_dir = str(tmp_path / "train" / "0"), str(tmp_path / "kodak")
    _png_dir(train_dir, 4, 80, 96, 5)
    _png_dir(eval_dir, 1, 64, 64, 6)
    os.makedirs(tmp_path / "checkpoint")
    cfg = tmp_path / "hific_d.yaml"
    cfg.write_text(f"""_base_: [{os.path.relpath(os.path.join(ROOT, 'config', 'crdr_stage_3.yaml'), str(tmp_path))}]
pretrained_weight_path: null
ckpt_root: {tmp_path}/checkpoint
hip_graphs: true
keep_discriminator: true
discriminator:
  _delete_: true
  type: HiFiCDiscriminator
  in_ch: 3
  out_ch: 1
  main_ch: 16
  use_sn: true
dataset:
  batch_size: 2
  train_dataset:
    root_dir: {tmp_path}/train
    name: openimage
    type: ImageDataset
    image_size: 64
    subset_list: [0]
  eval_dataset:
    root_dir: {eval_dir}
    name: Kodak
    type: ImageDataset
""")
    out = _run([os.path.join(ROOT, "scripts", "train.py"), str(cfg), "-d", "cuda:0", "-b", "2", "-ti", "8", "-s", "8", "-l", "2", "-e", "100",
                "-nw", "0"], cwd=str(tmp_path))
    model_dir = tmp_path / "checkpoint" / "hific_d" / "model"
    assert "comp_model_iter8.pth.tar" in os.listdir(model_dir), os.listdir(model_dir)
    import torch
    names = [n for n in os.listdir(model_dir) if n.startswith("discriminator")]
    assert names, os.listdir(model_dir)
    sd = torch.load(model_dir / names[0], map_location="cpu")
    keys = list(next(v for k, v in sd.items() if isinstance(v, dict)).keys())
    assert "model.0.weight_orig" in keys and "model.0.weight_u" in keys and "model.8.weight_v" in keys, keys[:8]


def test_eval_grid_on_demo_images(tmp_path):
    """Kodak (q, beta) grid harness on the three demo images with random weights: the reference's csv layout, one bpp per
    quality shared by both betas (the bitstream does not depend on beta), finite PSNR."""
    import csv
    from scripts import eval_grid
    out = tmp_path / "grid.csv"
    res = eval_grid.main(["--config_path", os.path.join(ROOT, "config", "crdr.yaml"), "--img_dir", os.path.join(ROOT, "demo_images"),
                          "--out_csv", str(out), "--qualities", "0.0", "2.25", "--betas", "3.84", "0.0"])
    assert res["rows"] == 4 and res["images"] == 3
    with open(out) as f:
        rows = list(csv.DictReader(f))
    assert list(rows[0].keys()) == eval_grid.COLUMNS
    assert [(float(r["quality"]), float(r["beta"])) for r in rows] == [(0.0, 3.84), (2.25, 3.84), (0.0, 0.0), (2.25, 0.0)]
    # (with freshly initialised InterpChAtt weights every rate level is the identity, so q does not move bpp here)
    assert rows[0]["bpp"] == rows[2]["bpp"] and rows[1]["bpp"] == rows[3]["bpp"] and float(rows[0]["bpp"]) > 0
    assert all(np.isfinite(float(r["PSNR"])) for r in rows) and all(r["dataset"] == "kodak" for r in rows)
