import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/kivi_oracle.c via ctypes).  Test infrastructure only."""
    from oracle import kivi_oracle
    kivi_oracle.build()
    return kivi_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _ratio_log_path():
    p = os.environ.get("KIVI_RATIO_LOG")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d, "gemv_ratios.log") if os.path.isdir(d) else None


@pytest.fixture(autouse=True)
def _record_gemv_ratios(request):
    """After every test: the worst ratio of each bar the test held a result to (tests/helpers.py: gemv_close, the BARE north_star bar,
    no ulp slack) goes to the ratio log -- $KIVI_RATIO_LOG, or gpurun_out/gemv_ratios.log on a GPU box -- so the margin is visible."""
    import helpers
    helpers.RATIOS.clear()
    yield
    path = _ratio_log_path()
    if path and helpers.RATIOS:
        worst = {}
        for rtol, ratio, n in helpers.RATIOS:
            w = worst.setdefault(rtol, [0.0, 0, 0])
            w[0] = max(w[0], ratio); w[1] += 1; w[2] += n
        with open(path, "a") as f:
            for rtol in sorted(worst):
                w = worst[rtol]
                f.write(f"{request.node.nodeid}\tbar={rtol}\tworst_ratio={w[0]:.4f}\tcalls={w[1]}\telements={w[2]}\n")
    helpers.RATIOS.clear()
