import os
import sys

import pytest

# the tests exercise the LPIPS arithmetic on seeded / random weights on purpose (no pretrained file exists offline)
os.environ.setdefault("CRDR_ALLOW_RANDOM_LPIPS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (stock torch on the host) is the slow half of every parity test, and torch's default -- one thread per physical core, 128 on
    # the GPU boxes, whose cores are shared between GPU slots -- is its WORST setting there: the stage-3 oracle step at N = 4 takes 26-30 s
    # at 128 threads, 9.7 s at 64, 5.2 s at 32, 3.5 s at 16 (tools/experiments/r6_oracle_threads.py).  (A different split of oneDNN's
    # reductions moves the oracle's values by rounding only, far below every tolerance the comparisons hold.)
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 16))


@pytest.fixture(scope="session")
def hip_lib():
    from crdr_amd.hip import lib
    return lib.load()


@pytest.fixture(autouse=True, scope="session")
def _record_forced_decisions():
    """every O.check_forced / O.check_forced_indexes of the session also reports its counts to tests/parity_margins.py"""
    from oracle import crdr_oracle as O
    from tests import parity_margins as PM
    cf, cfi = O.check_forced, O.check_forced_indexes

    def check_forced(report, numel):
        PM.record_forced(report)
        return cf(report, numel)

    def check_forced_indexes(report):
        PM.record_forced(report)
        return cfi(report)
    O.check_forced, O.check_forced_indexes = check_forced, check_forced_indexes
    yield
    O.check_forced, O.check_forced_indexes = cf, cfi


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("CRDR_PARITY_DUMP")
    if path:
        from tests import parity_margins
        parity_margins.dump(path)
