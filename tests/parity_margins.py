"""Measured parity margins of the GPU tests against the oracle, and the tolerances derived from them.

Every gradient / forward comparison of the parity tests reports its error here (`record`).  With CRDR_PARITY_DUMP=<path> in the
environment the session writes what it measured (per test, per parameter group: worst relative L2 error and the tensor that had
it) to that path; the committed copy is profiles/r5_parity_margins.json (r4_… / r3_… before).  `tolerance(group, cap)` then gates each comparison at
3 x the committed measurement (never looser than `cap`, the analytical bound the test states; `cap` alone when no measurement is
committed for that test / group).  fp32 MFMA is an exact fma chain and the tests run the library's built-in plans, so the
measured errors are reproducible run to run and box to box: a change that moves one by more than 3 x is a regression (or a new
summation order: re-measure with tools/parity_margins.sh and commit)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMITTED = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_parity_margins.json") for r in (6, 5, 4, 3)) if os.path.exists(p)),
                 os.path.join(ROOT, "profiles", "r6_parity_margins.json"))   # the newest committed measurement
FACTOR = 3.0
FLOOR = 1e-6   # a few fp32 ulps: comparisons that measured ~0 (bit-equal on the day) still get this much

_measured = {}
_committed = None


def _test_id() -> str:
    t = os.environ.get("PYTEST_CURRENT_TEST", "unknown")
    return t.rsplit(" (", 1)[0].split("::", 1)[-1]


def group_of(name: str) -> str:
    p = name.split(".")
    if p[0] == "context_model" and len(p) > 1:
        return ".".join(p[:2])
    if p[0] == "subD_list" and len(p) > 1:
        return ".".join(p[:2])
    return p[0]


def record(group: str, name: str, err: float) -> None:
    t = _measured.setdefault(_test_id(), {})
    g = t.setdefault(group, {"max": 0.0, "worst": None, "n": 0})
    g["n"] += 1
    if err >= g["max"]:
        g["max"], g["worst"] = float(err), name


def record_forced(report: dict) -> None:
    """Decisions the oracle adopted from the device inside its windows (oracle.FORCE_TOL around a rounding boundary, oracle.INDEX_TOL
    around a scale-table entry) in the current test: kept beside the margins so that a drift towards the 0.1 % bound is visible."""
    t = _measured.setdefault(_test_id(), {})
    f = t.setdefault("forced_decisions", {})
    for k in ("adopted", "mismatch", "symbols", "idx_adopted", "idx_mismatch", "indexes"):
        if k in report:
            f[k] = max(f.get(k, 0), int(report[k]))
    if f.get("symbols"):
        f["adopted_fraction"] = f.get("adopted", 0) / f["symbols"]


def record_imposed(report: dict) -> None:
    """ReLU-mask decisions the oracle adopted from the device inside oracle.MASK_WINDOW (the deterministic upstream-gradient gate)"""
    t = _measured.setdefault(_test_id(), {})
    t["imposed_relu_masks"] = {k: (float(v) if isinstance(v, float) else int(v)) for k, v in report.items()}
    if report.get("elements"):
        t["imposed_relu_masks"]["flipped_fraction"] = report.get("flipped", 0) / report["elements"]


def tolerance(group: str, cap: float) -> float:
    global _committed
    if os.environ.get("CRDR_PARITY_REMEASURE") == "1":   # tools/parity_margins.sh: gate at the stated bounds only, record afresh
        return cap
    if _committed is None:
        _committed = json.load(open(COMMITTED))["tests"] if os.path.exists(COMMITTED) else {}
    m = _committed.get(_test_id(), {}).get(group)
    if m is None:
        return cap
    return min(cap, max(FACTOR * m["max"], FLOOR))


def dump(path: str) -> None:
    if not _measured:
        return
    prev = {}
    if os.path.exists(path):
        prev = json.load(open(path)).get("tests", {})
    prev.update(_measured)
    with open(path, "w") as f:
        json.dump({"what": "worst relative L2 error (gradients) / max-abs error over the tensor scale (forward values) of the HIP path against "
                           "the CPU oracle, per test and parameter group; the tests gate at 3 x these (tests/parity_margins.py)",
                   "factor": FACTOR, "tests": prev}, f, indent=1, sort_keys=True)
