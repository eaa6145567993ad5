"""A few eager stage-3 steps at a fixed rate index for rocprofv3 --pmc passes (HBM traffic of the conv kernels).

The last step is the measured one: the script writes how many conv / wgrad launches one step makes (counted by the
library itself) to gpurun_out/pmc_step_meta.json, tools/pmc_traffic.py then takes that many trailing dispatches.
Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_step.py"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def main():
    q = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=False)
    ops.AUTOTUNE = True
    ops.load_tune_cache(ops.DEFAULT_TUNE_DB)
    loader = iter(tr.train_loader)
    lib = L.load()
    for it in range(1, 4):
        if it == 3:
            torch.cuda.synchronize()
            lib.crdr_profile_enable(1)
        tr.optimize_parameters(it, {**next(loader), "rate_ind": q})
    torch.cuda.synchronize()
    lib.crdr_profile_enable(0)
    out = {}
    for kind, name in ((0, "igemm"), (1, "wgrad"), (3, "winograd"), (4, "winograd_wgrad"), (5, "winograd_f4x4")):
        fl, ms, n = C.c_double(), C.c_double(), C.c_longlong()
        lib.crdr_profile_read(kind, C.byref(fl), C.byref(ms), C.byref(n))
        out[name] = {"launches": n.value, "gflop": fl.value / 1e9, "ms": ms.value}
    out["rate_ind"] = q
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pmc_step_meta.json", "w") as f:
        json.dump(out, f)
    print(out)


if __name__ == "__main__":
    main()
