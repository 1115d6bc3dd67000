"""Statistic mode end to end (densification inputs, litegs/utils/statistic_helper.py): fragment counts / weights / error moments
accumulated by the executor and by the operator path must agree with each other and with the oracle's statistic-mode blend, and
the heavy-tiles-first schedule of the second visit must not change the image."""
import numpy as np
import pytest
import torch

from tests.util import case, oracle_forward

pytestmark = pytest.mark.gpu


def _run(fused_path: bool, steps: int = 2):
    from litegs_amd import synthetic as S
    from litegs_amd.statistics import STATS
    from litegs_amd.trainer import SyntheticTrainer
    c = case("small")
    tr = SyntheticTrainer(c["n"], c["W"], c["H"], 300.0, n_frames=1, scene=c["params"], fused=fused_path, fuse_adam=False)
    # same camera as the oracle case
    from litegs_amd.trainer import Frame
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    tr.frames = [Frame(view, proj, planes, tr.frames[0].gt, 0)]
    for g in tr.opt.param_groups:
        g["lr"] = 0.0                                          # keep the cloud fixed: both visits see the same scene
    tr.sched = type("NoSchedule", (), {"step": lambda self: None})()
    STATS.reset(tr.n_chunks, tr.S, enabled_for_epoch=lambda e: True, device="cuda")
    STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()      # the singleton outlives other tests' trainers
    imgs = []
    with STATS.epoch(0):
        for i in range(steps):
            tr.step(0)
            imgs.append(tr.forward_only(0).cpu().numpy())
    torch.cuda.synchronize()
    out = {k: (m.sum.cpu().numpy().copy(), m.square_sum.cpu().numpy().copy(), m.count.cpu().numpy().copy()) for k, m in STATS.moments.items()}
    sched = {k: v.cpu().numpy().copy() for k, v in STATS.tile_schedule.items()}
    STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
    STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
    return out, imgs, sched


def test_statistic_mode_fused_equals_operator_path_and_oracle(oracle):
    res = oracle_forward("small", stat=True)
    c = case("small")
    so, imgs_o, sched_o = _run(False)
    sf, imgs_f, sched_f = _run(True)
    assert set(so) == set(sf) == {"fragment_weight", "fragment_err"}
    # the schedule (heavy tiles first) is a permutation of all tiles and leaves the image untouched
    ntiles = ((c["H"] + 7) // 8) * ((c["W"] + 15) // 16)
    for sched in (sched_o, sched_f):
        (t,) = sched.values()
        assert sorted(t.tolist()) == list(range(1, ntiles + 1))
    assert np.array_equal(imgs_o[0], imgs_o[1]) and np.array_equal(imgs_f[0], imgs_f[1]) and np.array_equal(imgs_o[0], imgs_f[0])
    for key in so:
        for a, b, what in zip(so[key], sf[key], ("sum", "square_sum", "count")):
            if what == "count":
                assert np.array_equal(a, b), f"{key}.{what}"
            else:
                scale = max(np.abs(a).max(), 1e-30)
                assert np.abs(a - b).max() / scale < 2e-5, f"{key}.{what}"
    # oracle: fragment counts / weight sums of ONE visit, scattered to the full cloud (two visits accumulated on the GPU)
    S = 128
    nvis = res.nvis
    cnt_full = np.zeros((c["params"][0].shape[-2], S), np.int64)
    w_full = np.zeros((c["params"][0].shape[-2], S), np.float64)
    cnt_full[res.visible_chunkid] = res.frag_count.reshape(nvis, S)
    w_full[res.visible_chunkid] = res.frag_weight.reshape(nvis, S)
    got_cnt = sf["fragment_weight"][2].reshape(cnt_full.shape)
    got_w = sf["fragment_weight"][0].reshape(w_full.shape)
    assert np.array_equal(got_cnt, 2 * cnt_full)
    assert np.abs(got_w - 2 * w_full).max() < 1e-3 * max(w_full.max(), 1e-9)


def test_executor_keeps_its_own_schedule_and_culling_after_a_statistics_epoch():
    """the reference rasterises along the statistics helper's cached tile list in every render after the first statistics epoch
    (litegs/render/__init__.py:75-79); the list is a permutation of all tiles, so the executor keeps its own schedule and depth-bound
    culling outside statistics renders: same image bit for bit, and the revisits really run culled"""
    from litegs_amd.statistics import STATS
    from litegs_amd.trainer import SyntheticTrainer
    tr = SyntheticTrainer(60000, 640, 360, 500.0, n_frames=2, seed=5)
    STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
    STATS.reset(tr.n_chunks, tr.S, enabled_for_epoch=lambda e: True, device="cuda")
    try:
        with STATS.epoch(0):
            for k in range(2):
                tr.forward_only(k)
        assert len(STATS.tile_schedule) == 2
        out = {}
        for always in (False, True):
            tr.renderer.stat_schedule_always = always
            torch.cuda.synchronize()                     # pinned feedback words of the previous renders have landed
            tr.renderer.reset_feedback()
            imgs, culled = [], False
            for visit in range(4):
                for k in range(2):
                    imgs.append(tr.forward_only(k).clone())
                    culled |= bool(tr.renderer.last_cull)
            torch.cuda.synchronize()
            out[always] = (imgs, culled)
        assert out[False][1] and not out[True][1]
        for a, b in zip(out[False][0], out[True][0]):
            assert torch.equal(a, b)
    finally:
        STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
        STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
