"""Per-kernel-family HBM traffic and achieved bandwidth of ONE eager stage-3 step, from the FETCH_SIZE and WRITE_SIZE passes of
tools/pmc_step.py (two separate `rocprofv3 --pmc` runs; counters cannot share a pass on gfx950):

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      (both counters are KiB; FETCH_SIZE tallies the 128-byte requests
                                                           of 16-byte-per-lane loads at 64 bytes, MI355X_MICROARCH.md, HBM)
The measured step is the LAST of the three identical steps the script runs: the dispatches between the 6th and the 9th
`adam_dyn_kernel` launch (three optimiser updates per step).  Durations are the dispatch timestamps of the FETCH pass.
Usage: python tools/pmc_families.py <fetch_dir> <write_dir> [out.json]"""
import csv
import glob
import json
import sys

FAMILIES = [("igemm_kernel", "igemm_kernel"), ("gemm1x1_kernel", "gemm1x1_kernel"), ("wino4_filter_batched", "wino4_filter_batched_kernel"), ("wino4_filter_kernel", "wino4_filter_kernel"), ("wino4_kernel", "wino4_kernel"), ("wino_filter_kernel", "wino_filter_kernel"), ("wino_wgrad_kernel", "wino_wgrad_kernel"), ("wino4_wgrad_kernel", "wino4_wgrad_kernel"), ("wino_kernel", "wino_kernel"), ("wgrad_kernel", "wgrad_kernel"), ("wgrad_reduce_batched", "wgrad_reduce_batched_kernel"),
            ("igemm_splitk_epilogue", "igemm_splitk_epilogue"), ("ebwd", "ebwd_kernel"), ("adam_dyn", "adam_dyn_kernel"),
            ("pack_weights_batched", "pack_weights_batched_kernel"), ("colsum_finish", "colsum_finish_"), ("colsum", "colsum_kernel"),
            ("gauss_cond_finish", "gauss_cond_finish_kernel"), ("gauss_cond_fwd", "gauss_cond_fwd_kernel"), ("gauss_cond_bwd", "gauss_cond_bwd_kernel"), ("eb", "eb_"),
            ("lrp", "lrp_"), ("col2im_rgb", "col2im_rgb_kernel"), ("lpips", "lpips_layer"), ("maxpool", "maxpool3s2"),
            ("reduce", "reduce_kernel"), ("aten", "at::native")]


def load(d, counter):
    import os
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # gpurun_out accumulates old runs
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "adam_dyn_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 9, f"expected three steps (9 optimiser launches), found {len(marks)}"
    return rows[marks[-4] + 1: marks[-1] + 1]


def family(name):
    for fam, pat in FAMILIES:
        if pat in name:
            return fam
    return "other"


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    assert len(fe) == len(wr), (len(fe), len(wr))
    acc = {}
    for a, b in zip(fe, wr):
        assert a["Kernel_Name"] == b["Kernel_Name"]
        fam = family(a["Kernel_Name"])
        t = acc.setdefault(fam, {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "ns": 0})
        t["launches"] += 1
        t["fetch_kib"] += float(a["Counter_Value"])
        t["write_kib"] += float(b["Counter_Value"])
        t["ns"] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from crdr_amd.hip import lib as L
    out = {"library_version": int(L.load().crdr_version()),
           "note": "one eager stage-3 step (bs 16, 256x256, q = 2); hbm = (2*FETCH_SIZE + WRITE_SIZE) KiB; GB/s over the kernels' own durations in the FETCH pass",
           "kernels_in_step": len(fe), "families": {}}
    for fam, t in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
        hbm = (2 * t["fetch_kib"] + t["write_kib"]) * 1024
        out["families"][fam] = {"launches": t["launches"], "ms": round(t["ns"] / 1e6, 3), "hbm_bytes": int(hbm),
                                "hbm_bytes_per_launch": int(hbm / t["launches"]), "achieved_GBps": round(hbm / max(t["ns"], 1), 1),
                                "frac_of_8TBps": round(hbm / max(t["ns"], 1) / 8000.0, 4)}
    # the family bench.py's roofline covers: tiled + streaming 1x1 forward / input-gradient launches
    both = [acc[k] for k in ("igemm_kernel", "gemm1x1_kernel", "wino_kernel", "wino_filter_kernel", "wino4_kernel", "wino4_filter_kernel", "wino4_filter_batched") if k in acc]
    if both:
        n = sum(acc[k]["launches"] for k in ("igemm_kernel", "gemm1x1_kernel", "wino_kernel", "wino4_kernel") if k in acc)   # (a Winograd launch = filter transform + kernel)
        hbm = sum((2 * t["fetch_kib"] + t["write_kib"]) * 1024 for t in both)
        ns = sum(t["ns"] for t in both)
        out["families"]["conv_fwd_dgrad"] = {"launches": n, "ms": round(ns / 1e6, 3), "hbm_bytes": int(hbm),
                                             "hbm_bytes_per_launch": int(hbm / n), "achieved_GBps": round(hbm / max(ns, 1), 1),
                                             "frac_of_8TBps": round(hbm / max(ns, 1) / 8000.0, 4),
                                             "note": "igemm_kernel + gemm1x1_kernel + wino_kernel + wino4_kernel (+ their filter transforms, the batched rebuild behind the optimiser updates included)"}
        # the same family without the batched filter rebuild behind the optimiser updates (bench.py reports both figures)
        nb = [acc[k] for k in ("igemm_kernel", "gemm1x1_kernel", "wino_kernel", "wino_filter_kernel", "wino4_kernel", "wino4_filter_kernel") if k in acc]
        hbm2, ns2 = sum((2 * t["fetch_kib"] + t["write_kib"]) * 1024 for t in nb), sum(t["ns"] for t in nb)
        out["families"]["conv_fwd_dgrad_no_rebuild"] = {"launches": n, "ms": round(ns2 / 1e6, 3), "hbm_bytes": int(hbm2), "hbm_bytes_per_launch": int(hbm2 / n),
                                                        "achieved_GBps": round(hbm2 / max(ns2, 1), 1), "frac_of_8TBps": round(hbm2 / max(ns2, 1) / 8000.0, 4),
                                                        "note": "as conv_fwd_dgrad, wino4_filter_batched left out"}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
