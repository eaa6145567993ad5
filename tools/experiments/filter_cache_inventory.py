"""What the persistent F(4x4) filter caches of a stage-3 trainer hold after warm-up: bytes per cache, caches per weight pack (duplicates of one
pack's transformed filters under different launch shapes / groupings are rebuilt separately by crdr_w4_filters_batched)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402

ops.AUTOTUNE = True
ops.load_tune_cache(ops.DEFAULT_TUNE_DB)
tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=False)
loader = iter(tr.train_loader)
for it in range(1, 7):
    d = next(loader)
    tr.optimize_parameters(it, {**d, "rate_ind": it % 5})
torch.cuda.synchronize()
rows, per_pack = [], {}
for key, e in ops._filter_cache.items():
    wk, G = key[0], key[1]
    rows.append({"G": G, "N": key[3], "H": key[4], "W": key[5], "C": key[6], "OC": key[9], "k": key[10], "stride": key[12], "transposed": key[14], "MB": round(e.nbytes / 1e6, 1),
                 "packs": len(set(wk))})
    for p in set(wk):
        per_pack.setdefault(p, []).append(e.nbytes / max(len(wk), 1))
tot = sum(r["MB"] for r in rows)
dup = sum(sum(v) - max(v) for v in per_pack.values()) / 1e6
out = {"caches": len(rows), "total_MB": round(tot, 1), "distinct_packs": len(per_pack), "MB_beyond_one_cache_per_pack": round(dup, 1),
       "largest": sorted(rows, key=lambda r: -r["MB"])[:25]}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r6_filter_cache_inventory.json", "w"), indent=1)
