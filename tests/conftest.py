import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "stochastic: the assertion bounds a quantity that float atomics make differ from run to run "
                                       "(training trajectories); collected behind every deterministic parity test")


# Collection order of the GPU suite (the driver runs `-m gpu -x`): deterministic operator / full-size parity against the oracle first,
# in the order of SURVEY.md section 8's rows, then the executor-level parity files, then everything whose assertion is a bound on a
# noisy training trajectory, diagnostics (test_zz_*) last.  A wobble in a noise-bounded test can then never hide a parity row.
_FILE_RANK = {name: i for i, name in enumerate((
    "test_gpu_ops.py", "test_gpu_reference_call_pattern.py", "test_gpu_fullsize.py", "test_gpu_pipeline.py", "test_gpu_fused.py", "test_gpu_tilesort.py", "test_gpu_tile_sort.py", "test_gpu_trained_cloud.py",
    "test_gpu_tilesizes.py", "test_gpu_edge.py", "test_gpu_stats.py", "test_gpu_schedule.py", "test_gpu_loss.py", "test_gpu_knn.py",
    "test_gpu_refine.py", "test_gpu_lifetime.py", "test_gpu_cull.py", "test_gpu_adam_skip.py", "test_gpu_dp.py", "test_gpu_training.py",
    "test_gpu_convergence.py", "test_bench_contract.py"))}


def pytest_collection_modifyitems(config, items):
    def key(it):
        fname = os.path.basename(str(it.fspath))
        if it.get_closest_marker("gpu") is None:
            return (0, 0, 0)                                   # CPU tests keep their (alphabetical) order, in front
        if fname.startswith("test_zz_"):
            return (3, 0, 0)
        noisy = 2 if it.get_closest_marker("stochastic") is not None else 1
        return (noisy, _FILE_RANK.get(fname, len(_FILE_RANK)), 0)
    items.sort(key=key)                                        # stable: the order inside a file is kept


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


@pytest.fixture(scope="session", autouse=True)
def _env_tuning():
    """LITEGS_TUNING=key=value[,...] (measurement / A-B aid): the launch variants of lg_set_tuning for the whole test process, so that a
    non-default kernel variant can be held to the same suite (`LITEGS_TUNING=28=1 pytest -m gpu`); operator tests do not import fast.py"""
    import os
    if os.environ.get("LITEGS_TUNING"):
        from litegs_amd import fast
        fast._apply_env_tuning()
    yield
