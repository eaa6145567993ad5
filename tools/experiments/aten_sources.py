"""Where do the non-crdr launches of a stage-3 step come from?  One eager step under torch.profiler with Python stacks; every device kernel that is
not one of the library's (at::native::*, rocclr copies / fills) is attributed to the innermost crdr_amd / bench frame of the op that launched it.
python tools/experiments/aten_sources.py [--all] > gpurun_out/aten_sources.txt"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true", help="attribute the library's own small launches too")
    a = ap.parse_args()
    from crdr_amd.hip import ops
    ops.AUTOTUNE = True
    ops.load_tune_cache(ops.DEFAULT_TUNE_DB)
    tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=False)
    loader = iter(tr.train_loader)
    levels = getattr(tr.comp_model, "rate_level", 0)

    def step(it):
        d = next(loader)
        if levels:
            d = {**d, "rate_ind": it % levels}
        tr.optimize_parameters(it, d)
    for it in range(1, 4):
        step(it)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(4)
        torch.cuda.synchronize()
    root = os.path.abspath(".")
    by_src = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    total = collections.Counter()
    for ev in prof.events():
        kerns = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
        if not kerns:
            continue
        src = None
        for fr in (ev.stack or []):
            if ("crdr_amd" in fr or "bench.py" in fr) and "profiler" not in fr:
                src = fr.replace(root + "/", "")
                break
        for k in kerns:
            lib_kernel = k.name.startswith("crdr::") or "crdr::" in k.name
            total["crdr" if lib_kernel else "other"] += 1
            if lib_kernel and not a.all:
                continue
            e = by_src[(src or "?", ev.name)]
            e[0] += 1
            e[1] += k.duration
            e[2][k.name[:60]] += 1
    print("launches in one eager step:", dict(total))
    for (src, op), (n, us, names) in sorted(by_src.items(), key=lambda kv: -kv[1][0]):
        print(f"{n:4d} launches {us:8.1f} us  {op:28s} {src}   [{', '.join(f'{c}x {nm}' for nm, c in names.most_common(2))}]")


if __name__ == "__main__":
    main()
