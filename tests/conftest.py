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
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(autouse=True)
def _statistics_singleton_is_left_clean(request):
    """litegs_amd.statistics.STATS is a process-wide singleton (as the reference's StatisticsHelper): a test that ran a statistics epoch must
    not hand its cached per-frame tile lists to the next test's renderers."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from litegs_amd.statistics import STATS
    STATS.active = False
    STATS.tile_schedule.clear()
    STATS.tile_blend_count.clear()


def pytest_terminal_summary(terminalreporter):
    """XFAIL reasons are findings of a diagnostic (test_zz_gpu_poison.py): print them even under -q"""
    for rep in terminalreporter.stats.get("xfailed", []):
        reason = getattr(rep, "wasxfail", "") or ""
        terminalreporter.write_line(f"XFAIL {rep.nodeid}: {reason[:3500]}")
