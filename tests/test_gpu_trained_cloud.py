"""Parity on a cloud that training produced, not on the synthetic test scenes: a student is trained for a few epochs with density control
(clone / split / prune, opacity decay, Morton re-sort -- the reference's loop, litegs/training/trainer.py:108-195), then every way the
executor builds its tile lists must reproduce, bit for bit, what the oracle's binning (get_allocate_size -> stable depth order ->
create_table -> tile_range) makes of the executor's OWN per-splat records, and the oracle's blend of that table must give the executor's
image (flip pins of this file are set by hand to 60: the trained cloud differs run to run -- float atomics -- so the observed count, 0-3
of 230 k pixels, is not a constant of the build).  tools/late_phase.py runs the same check on the epoch-120 cloud of the 3 M / 150-camera run (profiles/r04_late_phase_parity.log)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tables_on_a_density_controlled_cloud_match_the_oracle_in_every_route(oracle):
    from litegs_amd import densify as D, synthetic as S
    from litegs_amd._lib import lib
    from litegs_amd.trainer import SyntheticTrainer
    from tests.util import IMG_FLIP, assert_close
    n, W, H, f, frames = 60_000, 640, 360, 420.0, 4
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
    L = lib()
    rd = tr.renderer
    rd.stat_schedule_always = False                           # the executor's own path end to end
    for depth_order, scatter in ((0, True), (1, True), (1, False)):
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
            img_o, *_ = oracle.raster_forward(spt_o, ts_o, rec_o, H, W, 8, 16)
            assert_close(img.cpu().numpy(), np.clip(img_o[..., :H, :W], 0, 1), **IMG_FLIP, name=f"img route {depth_order}/{int(scatter)} frame {k}")
    tr.close()
