"""Parity on a cloud that training produced, not on the synthetic test scenes -- the state 99 % of a real run is in.  A student is trained
for a few epochs with density control (clone / split / prune, opacity decay, Morton re-sort -- the reference's loop,
litegs/training/trainer.py:108-195; the recipe of bench.py's `training_state`, shrunk), then

  * tables: every way the executor builds its tile lists must reproduce, bit for bit, what the oracle's binning (get_allocate_size ->
    stable depth order -> create_table -> tile_range) makes of the executor's OWN per-splat records, and the oracle's blend of that table
    must give the executor's image;
  * image AND all six parameter-gradient tensors of the whole path (projection -> binning -> blend forward / backward -> chain backward ->
    activation backward) against the oracle's pipeline on the same trained parameters, 1e-4 normalised, in each list route;
  * statistic mode: the three per-splat statistics the executor carries in slots 9-11 of the gradient record (fragment count, fragment
    weight sum: GR/raster.cu:283-302; err_square: GR/raster.cu:781-783) against the oracle's statistic-mode blend.

Flip pins of this file are set by hand (tests/golden/flip_pins.json: images 60, gradients 40): the trained cloud differs from run to run
-- float atomics in training -- so the observed counts are not constants of the build; every count is logged.  A trained cloud flips more
often than the synthetic one (profiles/r04_late_phase_parity.log: 87-209 of 6.2 M pixels at 3 M Gaussians against <= 24): opacity decay
and pruning-by-weight leave a large population of splats whose alpha sits near the 1/255 cut-off in many pixels, and every such (pixel,
splat) pair is one more place where a 1-ulp difference in exp() decides differently.  tools/late_phase.py runs the table check on the
epoch-120 cloud of the 3 M / 150-camera run."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


N0, WIDTH, HEIGHT, FOCAL, FRAMES = 60_000, 640, 360, 420.0, 4
ROUTES = ((0, True), (1, True), (1, False))          # (depth order, tile scatter): global route, tile scatter, tile radix sort


@pytest.fixture(scope="module")
def trained():
    """the density-controlled cloud (one training run per module) -> the trainer, flushed, optimisation frozen (lr 0, no schedule)"""
    from litegs_amd import densify as D, synthetic as S
    from litegs_amd.statistics import STATS
    from litegs_amd.trainer import SyntheticTrainer
    n, W, H, f, frames = N0, WIDTH, HEIGHT, FOCAL, FRAMES
    teacher_scene = S.make_scene(n, seed=3)
    teacher = SyntheticTrainer(n, W, H, f, n_frames=frames, seed=3, scene=teacher_scene, noise_targets=False)
    targets = [teacher.forward_only(k).clamp(0, 1).clone() for k in range(frames)]
    teacher.close()
    tr = SyntheticTrainer(n, W, H, f, n_frames=frames, seed=3, scene=S.perturb(teacher_scene, 4, amount=0.5), noise_targets=False)
    for k in range(frames):
        tr.frames[k].gt = targets[k]
    tr.speculative = True
    tr.enable_densify(D.DensifyParams(target_primitives=int(1.3 * n)), total_epochs=40, seed=3)
    for epoch in range(14):                                   # densifications at epochs 5 and 10, an opacity decay at 10, two re-sorts
        tr.degree = min(epoch // 5, 3)
        with tr.begin_epoch(epoch):
            for k in range(frames):
                tr.step(k)
        tr.end_epoch(epoch)
    tr.flush()
    assert tr.n_chunks * tr.S > n                             # the cloud really grew
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched = type("NoSchedule", (), {"step": lambda self: None})()
    STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
    STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
    yield tr
    tr.close()


def test_tables_on_a_density_controlled_cloud_match_the_oracle_in_every_route(oracle, trained):
    from litegs_amd._lib import lib
    from tests.util import assert_bracket, bracket_of
    tr, H, W = trained, HEIGHT, WIDTH
    L = lib()
    rd = tr.renderer
    rd.stat_schedule_always = False                           # the executor's own path end to end
    for depth_order, scatter in ROUTES:
        rd.depth_order, rd.tile_scatter = depth_order, scatter
        rd.reset_feedback()
        for k in (0, 2):
            with torch.no_grad():
                img = tr.forward_only(k)                      # first visit after the reset: unculled, exact sizes
            torch.cuda.synchronize()
            ws1, N = rd.last_ws1
            ws2, tl, _ = rd.last_ws2
            rec = ws1[L.lg_fused_packed_offset(N):][:4 * N * 16].view(torch.float32).view(N, 16).cpu().numpy()
            alloc_x = ws1[L.lg_fused_alloc_offset(N):][:4 * N].view(torch.int32).cpu().numpy()
            emitted = alloc_x > 0
            total = int(rd.fb_total[k])
            o_ts = L.lg_fused_tile_start_offset(tl, N, H, W, 8, 16)
            o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), tl, N, H, W, 8, 16)
            ts = ws2[o_ts:o_ts + 4 * (rd.ntiles + 2)].view(torch.int32).cpu().numpy()
            pts = ws2[o_pts:o_pts + 4 * total].view(torch.int32).cpu().numpy()
            ndc = np.zeros((1, 4, N), np.float32); ndc[0, 0] = np.where(emitted, rec[:, 13], 0); ndc[0, 1] = np.where(emitted, rec[:, 14], 0)
            vz = np.where(emitted, rec[:, 12], np.float32(3.0e38)).astype(np.float32)[None]
            inv = np.zeros((1, 2, 2, N), np.float32)
            inv[0, 0, 0] = np.where(emitted, rec[:, 9], 0); inv[0, 0, 1] = inv[0, 1, 0] = np.where(emitted, rec[:, 10], 0); inv[0, 1, 1] = np.where(emitted, rec[:, 11], 0)
            op = np.where(emitted, rec[:, 5], np.float32(0.0)).astype(np.float32)[None]
            _, _, alloc_o = oracle.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
            alloc_o = np.where(emitted, alloc_o[0], 0)
            assert np.array_equal(alloc_o, np.where(emitted, alloc_x, 0)), "tile counts"
            dsi = np.argsort(vz, axis=-1, kind="stable").astype(np.int64)
            prefix = np.cumsum(np.take_along_axis(alloc_o[None], dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
            assert int(prefix[0, -1]) == total
            st_o, spt_o, _, _ = oracle.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
            ts_o = oracle.tile_range(st_o, rd.ntiles)
            assert np.array_equal(ts, ts_o[0]), f"range table (depth order {depth_order}, scatter {scatter}, frame {k})"
            assert np.array_equal(pts, spt_o[0][:total]), f"lists (depth order {depth_order}, scatter {scatter}, frame {k})"
            rec_o = np.zeros((1, N, 16), np.float32)          # the oracle's record layout: px py a b c r g b opacity depth
            for dst, src in enumerate((0, 1, 9, 10, 11, 6, 7, 8, 5, 12)):
                rec_o[0, :, dst] = np.where(emitted, rec[:, src], np.float32(0.0))
            img_o, (img_lo, img_hi) = bracket_of(oracle, lambda: np.clip(oracle.raster_forward(spt_o, ts_o, rec_o, H, W, 8, 16)[0][..., :H, :W], 0, 1))
            assert_bracket(img.cpu().numpy(), img_o, [img_lo, img_hi], name=f"img route {depth_order}/{int(scatter)} frame {k}", decided_max=600)


def _host_params(tr):
    return [p.detach().cpu().numpy() for p in tr.params]


def _frame(tr, k):
    fr = tr.frames[k]
    return fr, [x.cpu().numpy() for x in (fr.view, fr.proj, fr.planes)]


def _aabb(tr):
    """chunk bounding boxes of the CURRENT parameters (the trainer's own are those of the last re-sort, as in the reference: the cloud has
    moved since), which is what the oracle's pipeline derives from the parameters it is given"""
    from litegs_amd import render as R
    with torch.no_grad():
        return R.get_cluster_AABB(tr.params[0], tr.params[1].exp(), torch.nn.functional.normalize(tr.params[2], dim=0))


@pytest.mark.parametrize("route", range(len(ROUTES)), ids=["global", "tile_scatter", "tile_radix"])
def test_image_and_all_gradients_on_the_trained_cloud_match_the_oracle(oracle, trained, route):
    """the whole differentiable path on the trained parameters (what test_gpu_fullsize checks on the synthetic cloud)"""
    from litegs_amd import fast
    from tests.util import compacted_grads, parity_image_and_gradients
    tr, H, W = trained, HEIGHT, WIDTH
    depth_order, scatter = ROUTES[route]
    params_host = _host_params(tr)
    degree = int(tr.degree)
    origin, extend = _aabb(tr)
    for k in (1, 3):
        fr, (view, proj, planes) = _frame(tr, k)
        res = oracle.render_forward(params_host, view, proj, planes, H, W, degree)
        rd = fast.FusedRenderer(1, H, W)
        rd.depth_order, rd.tile_scatter = depth_order, scatter
        cam = fast.CameraFrame(fr.view, fr.proj, fr.planes, 0)
        for p in tr.params:
            p.grad = None
        rng = np.random.default_rng(11 + k)
        w_host = rng.standard_normal((1, 3, H, W)).astype(np.float32)
        img, vis_id, vis_num = rd.render(cam, origin, extend, *tr.params, degree)
        (img * torch.from_numpy(w_host).cuda()).sum().backward()
        torch.cuda.synchronize()
        assert int(vis_num.item()) == res.nvis
        assert np.array_equal(vis_id.cpu().numpy()[:res.nvis], res.visible_chunkid)
        assert abs(int(rd.fb_total[0]) - res.n_instances) <= max(2, int(2e-6 * res.n_instances)), (int(rd.fb_total[0]), res.n_instances)
        like = oracle.render_backward(res, params_host, view, proj, np.zeros_like(res.img), H, W, degree)[0]
        parity_image_and_gradients(oracle, res, img.detach().cpu().numpy(), compacted_grads(tr.params, res.nvis, like), params_host, view, proj,
                                   w_host, H, W, degree, tag=f" frame {k}", decided_max_img=600, decided_max_grad=200)
        rd.close()
    for p in tr.params:
        p.grad = None


@pytest.mark.parametrize("shift", [6, 9], ids=["segments_of_64", "segments_of_512"])
def test_segmented_blend_backward_matches_the_oracle_and_the_plain_walk(oracle, trained, shift):
    """lg_set_tuning(22, 1) (off by default: csrc/raster.hip g_bwd_segments): renders along a tile list take the lean blend forward, which then
    leaves a checkpoint of every pixel's {T, C} each 2^shift list positions, and a blend backward of one wave per (tile, segment)
    (csrc/raster.hip LG_UNIT_CLASSES, raster_backward_fast_kernel<.., SEG>).  Image and the six parameter gradients against the oracle
    under the suite's rule, and against the same frame with the segments switched off."""
    from litegs_amd import fast
    from litegs_amd._lib import check, lib
    from litegs_amd.statistics import STATS
    from tests.util import compacted_grads, parity_image_and_gradients
    tr, H, W = trained, HEIGHT, WIDTH
    L = lib()
    params_host = _host_params(tr)
    degree = int(tr.degree)
    origin, extend = _aabb(tr)
    k = 1
    fr, (view, proj, planes) = _frame(tr, k)
    res = oracle.render_forward(params_host, view, proj, planes, H, W, degree)
    rng = np.random.default_rng(23)
    w_host = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    like = oracle.render_backward(res, params_host, view, proj, np.zeros_like(res.img), H, W, degree)[0]
    grads = {}
    try:
        for segments in (1, 0):
            check(L.lg_set_tuning(22, segments), "tuning 22"); check(L.lg_set_tuning(23, shift), "tuning 23")
            rd = fast.FusedRenderer(1, H, W)
            cam = fast.CameraFrame(fr.view, fr.proj, fr.planes, 0)
            STATS.current_frame = 0
            STATS.tile_schedule[0] = torch.randperm(rd.ntiles, generator=torch.Generator().manual_seed(5)).to(torch.int32).cuda() + 1
            for p in tr.params:
                p.grad = None
            img, vis_id, vis_num = rd.render(cam, origin, extend, *tr.params, degree)
            (img * torch.from_numpy(w_host).cuda()).sum().backward()
            torch.cuda.synchronize()
            ws2, tl, N = rd.last_ws2
            if segments:
                counts = ws2[L.lg_fused_unit_count_offset(tl, N, H, W, 8, 16):][:4 * 17].view(torch.int32).cpu().numpy()      # full segments, 16 length classes
                units = int(counts.sum())
                print(f"[segments] shift {shift}: {units} units for {rd.ntiles} tiles")
                assert units > 0
                if shift == 6:
                    assert units > rd.ntiles, "lists of this cloud are longer than 64 entries: there must be more units than tiles"
            grads[segments] = [g.copy() for g in compacted_grads(tr.params, res.nvis, like)]
            if segments:
                parity_image_and_gradients(oracle, res, img.detach().cpu().numpy(), grads[1], params_host, view, proj, w_host, H, W, degree,
                                           tag=f" segments 2^{shift}", decided_max_img=600, decided_max_grad=200)
            rd.close()
    finally:
        check(L.lg_set_tuning(22, 0), "tuning 22"); check(L.lg_set_tuning(23, 9), "tuning 23")      # the defaults
        STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
        STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
        for p in tr.params:
            p.grad = None
    # the two walks differ by rounding only (T from the checkpoint instead of by divisions, Bd from a colour difference) and by the order of
    # the float atomics
    for name, a, b in zip(("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"), grads[1], grads[0]):
        scale = max(float(np.abs(b).max()), 1e-30)
        err = float(np.abs(a - b).max()) / scale
        print(f"[segments] {name}: max difference to the plain walk {err:.3e} of the largest element")
        assert err <= 2e-5, (name, err)


def test_statistics_in_the_gradient_record_match_the_oracle_on_the_trained_cloud(oracle, trained):
    """statistic epochs of the executor: fragment count, fragment weight sum and err_square travel in slots 9-11 of the blend backward's
    gradient record (csrc/raster.hip STAT == 2) and reach the statistics helper through lg_stat_accumulate -- against the oracle's
    statistic-mode blend forward (GR/raster.cu:283-302) and backward (GR/raster.cu:781-783) of the same frame"""
    from litegs_amd import fast
    from litegs_amd.statistics import STATS
    tr, H, W = trained, HEIGHT, WIDTH
    params_host = _host_params(tr)
    degree = int(tr.degree)
    k = 2
    fr, (view, proj, planes) = _frame(tr, k)
    res = oracle.render_forward(params_host, view, proj, planes, H, W, degree, enable_stat=True)
    rng = np.random.default_rng(5)
    w_host = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    d_img = np.zeros_like(res.img)
    inside = (res.img[..., :H, :W] >= 0) & (res.img[..., :H, :W] <= 1)
    d_img[..., :H, :W] = w_host * inside
    _, _, _, d_opa, esq = oracle.raster_backward(res.sorted_point, res.tile_start, res.packed, res.trans, res.last, d_img, H, W, 8, 16, enable_stat=True)
    S, C = tr.S, tr.n_chunks
    full = lambda x, dt: np.zeros((C, S), dt)
    cnt_o, w_o, err_o, esq_o = full(0, np.int64), full(0, np.float64), full(0, np.float64), full(0, np.float64)
    cnt_o[res.visible_chunkid] = res.frag_count.reshape(res.nvis, S)
    w_o[res.visible_chunkid] = res.frag_weight.reshape(res.nvis, S)
    err_o[res.visible_chunkid] = d_opa.reshape(res.nvis, S)
    esq_o[res.visible_chunkid] = esq.reshape(res.nvis, S)
    rd = fast.FusedRenderer(1, H, W)
    cam = fast.CameraFrame(fr.view, fr.proj, fr.planes, 0)
    origin, extend = _aabb(tr)
    STATS.reset(C, S, enabled_for_epoch=lambda e: True, device="cuda")
    STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
    STATS.current_frame = 0
    try:
        with STATS.epoch(0):
            img, _, _ = rd.render(cam, origin, extend, *tr.params, degree)
            (img * torch.from_numpy(w_host).cuda()).sum().backward()
        torch.cuda.synchronize()
        mw, me = STATS.moments["fragment_weight"], STATS.moments["fragment_err"]
        got_cnt = mw.count.cpu().numpy().reshape(C, S).astype(np.int64)
        got_w = mw.sum.cpu().numpy().reshape(C, S).astype(np.float64)
        got_w2 = mw.square_sum.cpu().numpy().reshape(C, S).astype(np.float64)
        got_err = me.sum.cpu().numpy().reshape(C, S).astype(np.float64)
        got_esq = me.square_sum.cpu().numpy().reshape(C, S).astype(np.float64)
        assert np.array_equal(me.count.cpu().numpy().reshape(C, S), mw.count.cpu().numpy().reshape(C, S))
    finally:
        STATS.reset(1, 1, enabled_for_epoch=lambda e: False, device="cuda")
        STATS.tile_schedule.clear(); STATS.tile_blend_count.clear()
        rd.close()
        for p in tr.params:
            p.grad = None
    # fragment counts are integers decided by alpha >= 1/255 per (pixel, splat): a threshold flip moves a count by one
    dc = np.abs(got_cnt - cnt_o)
    assert dc.max() <= 2 and (dc > 0).sum() <= max(40, int(2e-4 * (cnt_o > 0).sum())), (int(dc.max()), int((dc > 0).sum()), int((cnt_o > 0).sum()))
    assert cnt_o.sum() > 0 and abs(int(got_cnt.sum()) - int(cnt_o.sum())) <= 60
    for name, got, ref in (("weight", got_w, w_o), ("weight^2", got_w2, w_o * w_o), ("err", got_err, err_o), ("err_square", got_esq, esq_o)):
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(got - ref) / scale
        bad = int((err > 1e-4).sum())
        print(f"[stat] {name}: max normalised err {err.max():.3e}, {bad} beyond 1e-4 of {(ref != 0).sum()} non-zero")
        assert bad <= 40 and err.max() <= 5e-3, (name, bad, float(err.max()))
