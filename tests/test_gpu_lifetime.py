"""Several renderers in one process: the C executor keeps no process-wide state (LgFusedCtx), and the pinned words the device stores into
come from the library's arena, which is never unmapped (litegs_amd/hostwords.py).  Regression tests for the round-3 fault pattern -- "a
pointer that outlives a trainer": a process's later trainers, after ``empty_cache()`` has unmapped an earlier trainer's memory."""
import ctypes
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _steps(tr, n, start=0):
    return [float(tr.step((start + i) % len(tr.frames)).detach()) for i in range(n)]


@pytest.mark.stochastic
def test_two_trainers_and_an_evaluator_interleaved_across_empty_cache():
    """Trainer A (speculative culling, gradient replicas), trainer B (the same) and an evaluation renderer (no gradients, gated repeat) take
    turns; A is dropped WITHOUT a flush while its last steps are in flight, the caches are emptied, B and the evaluator go on and a third
    trainer is created on the recycled memory.  B's loss curve must be the one B produces alone (every trainer owns its hot counter,
    poison words and feedback words: nothing is inherited through the library)."""
    from litegs_amd import fast
    from litegs_amd.trainer import SyntheticTrainer
    N, W, H, F = 150_000, 640, 360, 380.0          # a camera inside the cloud: splats that cover >= 128 tiles exist (replicas are used)

    def make(seed):
        tr = SyntheticTrainer(N, W, H, F, n_frames=3, seed=seed)
        tr.speculative = True
        return tr

    solo = make(2)
    ref = _steps(solo, 36)
    solo.flush()
    solo.close()
    del solo

    a, b = make(1), make(2)
    ev = fast.FusedRenderer(3, H, W)               # evaluation: its frames have their own feedback words and depth bounds

    def evaluate(tr, k):
        with torch.no_grad():
            img, _, _ = ev.render(tr.frames[k].cam, tr.cluster_origin, tr.cluster_extend, *tr.params, tr.degree)
        return img

    got = []
    for r in range(6):                             # 6 rounds x (3 steps of A, 3 of B, one evaluation render of each)
        _steps(a, 3, 3 * r)
        got += _steps(b, 3, 3 * r)
        evaluate(a, r % 3); evaluate(b, r % 3)
    assert a.renderer.hot_counter is not None and b.renderer.hot_counter is not None
    assert a.renderer.hot_counter.data_ptr() != b.renderer.hot_counter.data_ptr()
    words_a = a.renderer._words.ptr
    _steps(a, 2)                                   # in flight ...
    del a                                          # ... and gone: no flush, no close
    gc.collect()
    torch.cuda.empty_cache()
    for r in range(6, 9):
        got += _steps(b, 3, 3 * r)
        evaluate(b, r % 3)
    c = make(3)                                    # lands on the memory A gave back
    for r in range(9, 12):
        _steps(c, 3, 3 * r)
        got += _steps(b, 3, 3 * r)
    b.flush(); c.flush()
    torch.cuda.synchronize()
    assert c.renderer._words.ptr != 0 and words_a != 0
    assert all(bool(torch.isfinite(p).all()) for p in b.params + c.params)
    # the first steps are the same computation bit for bit up to the blend backward's atomics order; later ones drift like two runs of one trainer
    np.testing.assert_allclose(got[:6], ref[:6], rtol=2e-4)
    np.testing.assert_allclose(got, ref, rtol=3e-2)
    b.close(); c.close(); ev.close()


def test_host_words_are_quarantined_until_a_device_sync():
    from litegs_amd.hostwords import HostWords
    w = HostWords(5)
    assert w.n == 5 and list(w.a) == [0] * 5
    w.a[:] = 7
    p = w.ptr
    x = torch.ones((1 << 20,), device="cuda")      # something in flight
    (x * 2).sum()
    w.close()
    assert w.ptr == 0
    w2 = HostWords(5)                              # freed ranges come back only behind a device synchronisation, zeroed
    assert list(w2.a) == [0] * 5
    assert w2.ptr % 64 == 0 and p % 64 == 0        # 64-byte granules: two owners never share a cache line
    t = w2.tensor()
    t[1] = 3
    assert int(w2.a[1]) == 3
    w2.close()


def test_executor_context_is_per_call():
    """the same scene through two renderers with different options in alternation: each one's tables are what it would build alone"""
    from litegs_amd import fast, render as R
    from litegs_amd._lib import lib
    from tests.util import case, oracle_forward
    L = lib()
    c = case("small")
    res = oracle_forward("small")
    H, W = c["H"], c["W"]
    params = [torch.from_numpy(p).cuda() for p in c["params"]]
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
    rds = []
    for mode, scatter in ((0, True), (1, True), (1, False)):
        rd = fast.FusedRenderer(1, H, W)
        rd.depth_order, rd.tile_scatter = mode, scatter
        rds.append(rd)
    cam = fast.CameraFrame(view, proj, planes, 0)
    imgs = []
    for _ in range(3):                             # first visit, culled revisits
        for rd in rds:
            with torch.no_grad():
                imgs.append(rd.render(cam, origin, extend, *params, c["degree"])[0].clone())
            if not rd.last_cull:
                torch.cuda.synchronize()
                ws2, table_len, N = rd.last_ws2
                o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), table_len, N, H, W, 8, 16)
                pts = ws2[o_pts:o_pts + 4 * res.n_instances].view(torch.int32).cpu().numpy()
                np.testing.assert_array_equal(pts, res.sorted_point[0])
    for img in imgs[1:]:
        assert torch.equal(img, imgs[0])


def test_a_foreign_cached_tile_list_is_not_this_frames_schedule():
    """the statistics helper's per-frame tile lists are keyed by frame index in a process-wide singleton: a renderer at another resolution
    must ignore a list another trainer cached under the same index (it used to rasterise along it: most tiles never written)"""
    from litegs_amd import fast, render as R
    from litegs_amd.statistics import STATS
    from tests.util import case
    c = case("small")
    H, W = c["H"], c["W"]
    params = [torch.from_numpy(p).cuda() for p in c["params"]]
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
    cam = fast.CameraFrame(view, proj, planes, 0)
    with torch.no_grad():
        ref = fast.FusedRenderer(1, H, W).render(cam, origin, extend, *params, c["degree"])[0].clone()
    STATS.current_frame = 0
    STATS.tile_schedule[0] = torch.arange(1, 41, dtype=torch.int32, device="cuda")          # 40 tiles of some other image
    try:
        with torch.no_grad():
            img = fast.FusedRenderer(1, H, W).render(cam, origin, extend, *params, c["degree"])[0]
        assert torch.equal(img, ref)
    finally:
        STATS.tile_schedule.clear()
