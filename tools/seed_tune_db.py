"""Seed for a perf-database rebuild: keep the entries of an older database whose kernels did not change, under the current
signature; the rest is tuned live by `bench.py --tune-db SEED --save-tune-db OUT`.

    python tools/seed_tune_db.py OLD.json SEED.json --signature v300-c27-s8-w23 [--min-rows 20000]

Conv entries ("c" / "g" / "m" keys) are kept only when the GEMM has at least --min-rows rows (pixels): below that the plan
space changed (split-K reduces inside the launch since v300); weight-gradient entries ("w*") are always kept."""
import argparse
import ast
import json


def rows(key):
    kind = key[0]
    if kind == "c":     # ("c", n, h, w, C, oh, ow, oc, k, stride, pad, transposed, ...)
        n, h, w, oh, ow, tr = key[1], key[2], key[3], key[5], key[6], key[11]
        return n * (h * w if tr else oh * ow)
    if kind == "g":     # ("g", G, n, h, w, ...)
        return key[2] * key[3] * key[4]
    if kind == "m":     # ("m", G, n, h, w, oh, ow, ...)
        return key[2] * min(key[3] * key[4], key[5] * key[6])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("out")
    ap.add_argument("--signature", required=True)
    ap.add_argument("--min-rows", type=int, default=20000)
    a = ap.parse_args()
    db = json.load(open(a.old))
    keep = {}
    for k, v in db["algos"].items():
        key = ast.literal_eval(k)
        r = rows(key)
        if r is None or r >= a.min_rows:
            keep[k] = v
    json.dump({"signature": a.signature, "algos": keep}, open(a.out, "w"), indent=0)
    print(f"kept {len(keep)} of {len(db['algos'])} entries")


if __name__ == "__main__":
    main()
