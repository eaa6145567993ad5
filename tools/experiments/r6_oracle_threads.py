"""How long the oracle's stage-3 step (the CPU half of the full-size parity tests) takes on this host at different torch thread counts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402,F401
from oracle import crdr_oracle as O  # noqa: E402
from crdr_amd.models import build_comp_model  # noqa: E402
from crdr_amd.models.discriminator import build_discriminator  # noqa: E402
from crdr_amd.losses.perceptual_loss import LpipsAlex  # noqa: E402
from crdr_amd.utils.options import BaseConfig, ConfigDict  # noqa: E402

cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(bench.ROOT, "config", "crdr_stage_3.yaml"))
cfg["device"] = "cpu"
torch.manual_seed(0)
g = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in build_comp_model(ConfigDict(cfg)).state_dict().items() if v.numel() > 0}
lp = {"lpips." + k: v.detach() for k, v in LpipsAlex().state_dict().items()}
d = {k: v.detach().clone().requires_grad_(True) for k, v in build_discriminator(ConfigDict(cfg).discriminator).state_dict().items()}
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 256
x = torch.rand(n, 3, size, size) * 2 - 1
ny, nz = torch.rand(n, 320, size // 16, size // 16) - 0.5, torch.rand(n, 192, size // 64, size // 64) - 0.5
print("default threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)
for th in [int(v) for v in os.environ.get("THREADS", "0,16,32,64,128").split(",")]:
    if th:
        torch.set_num_threads(th)
    t0 = time.perf_counter()
    losses, out = O.stage3_g_losses(g, d, lp, x, 2, 2.56, ny, nz)
    losses["total"].backward()
    print(f"threads {torch.get_num_threads():4d}: G losses + backward at N = {n}: {time.perf_counter() - t0:.1f} s", flush=True)
