"""Where do the ATen kernels of a stage-3 step come from?  One eager step under torch.profiler with Python stacks; device-side
ATen ops (fill, add, copy, mul, cat, mean ...) are grouped by the innermost frame inside this repository.
Usage: python tools/aten_sources.py [rate_ind]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def main():
    q = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=False)
    ops.AUTOTUNE = True
    ops.load_tune_cache(ops.DEFAULT_TUNE_DB)
    loader = iter(tr.train_loader)
    for it in range(1, 3):
        tr.optimize_parameters(it, {**next(loader), "rate_ind": q})
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr.optimize_parameters(3, {**next(loader), "rate_ind": q})
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.Counter()
    for e in prof.events():
        if not e.name.startswith("aten::"):
            continue
        kern = [k for k in (getattr(e, "kernels", None) or [])]
        if not kern:
            continue  # ATen ops that launched device work themselves
        where = "<autograd engine / no repo frame>"
        for fr in e.stack or []:
            if root in fr or "crdr_amd" in fr:
                where = fr.replace(root + "/", "")
                break
        if where.startswith("<"):  # no Python stack on this build: the operand shapes identify the site
            where = str([tuple(sh) for sh in (e.input_shapes or []) if sh])[:120]
        agg[(e.name, where)] += 1
    tot = sum(agg.values())
    print(f"{tot} device-side leaf ATen ops in one step")
    for (name, where), n in agg.most_common(60):
        print(f"{n:5d}  {name:28s} {where[:150]}")


if __name__ == "__main__":
    main()
