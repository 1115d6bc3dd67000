"""Edge cases of the path, fused executor and operator path side by side against the CPU oracle: nothing visible, nothing
rasterisable, images smaller than one tile, a single chunk, ragged chunk counts."""
import numpy as np
import pytest
import torch

from litegs_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _run_both(scene, view, proj, planes, H, W, degree=3, backward=True, ops_must_raise=False):
    """-> [(img, grads, n_vis) for the fused executor, same for the operator path].  ops_must_raise: frames without a single tile
    instance make the reference's create_table fail its TORCH_CHECK("error pred_allocate_size", GR/binning.cu:164); the operator
    surface keeps that behaviour (RuntimeError), the executor renders the background."""
    from litegs_amd import fast, render as R
    outs = []
    for mode in ("fused", "ops"):
        params = [torch.nn.Parameter(torch.from_numpy(p).cuda()) for p in scene]
        v, pj, pl = [torch.from_numpy(x).cuda() for x in (view, proj, planes)]
        with torch.no_grad():
            origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
        if mode == "fused":
            rd = fast.FusedRenderer(1, H, W)
            img, vis_id, vis_num = rd.render(fast.CameraFrame(v, pj, pl, 0), origin, extend, *params, degree)
            n_inst = None
        else:
            pp = R.PipelineParams()
            vis_id, vis_num, xyz, scale, rot, color, opacity = R.render_preprocess(origin, extend, pl, v, *params, None, None, pp, degree)
            if ops_must_raise:
                with pytest.raises(RuntimeError, match="pred_allocate_size"):
                    R.render(v, pj, xyz, scale, rot, color, opacity, vis_num * pp.cluster_size, None, None, degree, (H, W), pp)
                outs.append((None, None, int(vis_num.item())))
                continue
            img, *_ = R.render(v, pj, xyz, scale, rot, color, opacity, vis_num * pp.cluster_size, None, None, degree, (H, W), pp)
        if backward:
            img.sum().backward()
        torch.cuda.synchronize()
        grads = [None if p.grad is None else p.grad.compacted_values.detach().cpu().numpy() for p in params]
        outs.append((img.detach().cpu().numpy(), grads, int(vis_num.item())))
    return outs


def test_camera_looking_away_sees_nothing():
    """No chunk passes the frustum test: zero image, zero-sized work, finite (zero) gradients, no crash in either path."""
    scene = S.make_scene(3000, seed=1)
    view, proj, planes = S.make_camera(160, 96, 150.0, 150.0, (0.0, 0.0, 30.0), target=(0.0, 0.0, 60.0))     # cloud is behind the camera
    (img_f, g_f, nv_f), (_, _, nv_o) = _run_both(scene, view, proj, planes, 96, 160, ops_must_raise=True)
    assert nv_f == 0 and nv_o == 0
    assert np.all(img_f == 0)
    # compact gradients exist (one allocated chunk) but hold no valid row: everything >= visible_chunks_num is dirty by design


def test_transparent_cloud_emits_no_instances(oracle):
    """Every Gaussian fails the 1/255 opacity test (binning.cu:346): chunks are visible, the table is empty."""
    scene = list(S.make_scene(2500, seed=2))
    scene[5] = np.full_like(scene[5], -12.0)                      # sigmoid(-12) = 6e-6 < 1/255
    view, proj, planes = S.make_camera(200, 120, 180.0, 180.0, (1.8, -0.3, 0.8))
    ref = oracle.render_forward(scene, view, proj, planes, 120, 200, 3)
    assert ref.n_instances == 0 and ref.nvis > 0
    (img_f, g_f, nv_f), (_, _, nv_o) = _run_both(scene, view, proj, planes, 120, 200, ops_must_raise=True)
    assert nv_f == ref.nvis and nv_o == ref.nvis
    assert np.all(img_f == 0)
    for g in g_f:
        valid = g.reshape(-1, g.shape[-2], g.shape[-1])[:, :ref.nvis]
        assert np.isfinite(valid).all() and np.all(valid == 0)


@pytest.mark.parametrize("W,H", [(5, 3), (16, 8), (17, 9)])
def test_images_around_one_tile(oracle, W, H):
    """Image smaller than / equal to / just over one 8x16 tile (padded tiles, 1-2 tile grids)."""
    scene = S.make_scene(1500, seed=3, scale_mult=1.5)
    view, proj, planes = S.make_camera(W, H, 12.0, 12.0, (1.5, -0.2, 0.6))
    ref = oracle.render_forward(scene, view, proj, planes, H, W, 3)
    ref_img = np.clip(ref.img[..., :H, :W], 0, 1)
    (img_f, _, nv_f), (img_o, _, nv_o) = _run_both(scene, view, proj, planes, H, W, backward=True)
    assert nv_f == ref.nvis == nv_o
    assert np.array_equal(img_f, img_o)
    assert np.abs(img_f - ref_img).max() < 2e-2 and (np.abs(img_f - ref_img) > 1e-4).mean() < 1e-3


@pytest.mark.parametrize("n", [1, 128, 129, 1000])
def test_ragged_gaussian_counts(oracle, n):
    """1 Gaussian, exactly one chunk, one chunk + 1, a count that is not a multiple of the chunk size (padding Gaussians)."""
    scene = S.make_scene(n, seed=4, scale_mult=2.0)
    view, proj, planes = S.make_camera(96, 64, 80.0, 80.0, (1.6, -0.3, 0.7))
    ref = oracle.render_forward(scene, view, proj, planes, 64, 96, 3)
    ref_img = np.clip(ref.img[..., :64, :96], 0, 1)
    empty = ref.n_instances == 0
    (img_f, g_f, nv_f), (img_o, g_o, nv_o) = _run_both(scene, view, proj, planes, 64, 96, ops_must_raise=empty)
    assert nv_f == ref.nvis == nv_o
    if empty:
        assert np.all(img_f == 0)
        return
    assert np.array_equal(img_f, img_o)
    assert np.abs(img_f - ref_img).max() < 2e-2 and (np.abs(img_f - ref_img) > 1e-4).mean() < 1e-3
    assert all(np.isfinite(g.reshape(-1, g.shape[-2], g.shape[-1])[:, :ref.nvis]).all() for g in g_f + g_o)


def test_8k_image_with_giant_splats(oracle):
    """7680x4320: 259 200 tiles (> 16-bit staging keys, 19 key bits = 3 radix passes) and near-camera splats whose tile rectangle has
    more than 256 slices (the serial fallback of the cooperative emission); fused executor vs oracle, tables included."""
    from litegs_amd import fast, render as R
    H, W = 4320, 7680
    scene = list(S.make_scene(600, seed=9, scale_mult=3.0))
    view, proj, planes = S.make_camera(W, H, 5000.0, 5000.0, (1.2, -0.2, 0.5))
    ref = oracle.render_forward(scene, view, proj, planes, H, W, 3)
    assert ref.n_instances > 1_000_000 and ref.alloc.max() > 70_000, (ref.n_instances, int(ref.alloc.max()))
    params = [torch.nn.Parameter(torch.from_numpy(p).cuda()) for p in scene]
    v, pj, pl = [torch.from_numpy(x).cuda() for x in (view, proj, planes)]
    with torch.no_grad():
        origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
    rd = fast.FusedRenderer(1, H, W)
    img, vis_id, vis_num = rd.render(fast.CameraFrame(v, pj, pl, 0), origin, extend, *params, 3)
    img.sum().backward()
    torch.cuda.synchronize()
    assert int(vis_num.item()) == ref.nvis
    assert abs(int(rd.fb_total[0]) - ref.n_instances) <= max(2, int(2e-6 * ref.n_instances))
    err = np.abs(img.detach().cpu().numpy() - np.clip(ref.img[..., :H, :W], 0, 1))
    assert err.max() < 2e-2 and (err > 1e-4).mean() < 5e-5
    assert all(torch.isfinite(p.grad.compacted_values[..., :ref.nvis, :]).all() for p in params)


def test_giant_splats_are_emitted_in_parts(oracle):
    """1080p with near-camera splats of 1 000..16 000 tiles: the second emission launch splits every splat of more than 1024 tiles
    into parts handled by different waves; the table (operator path, exact and truncated length) stays bit-exact."""
    from litegs_amd import fused as F
    H, W = 1080, 1920
    scene = list(S.make_scene(2048, seed=21, scale_mult=1.5))
    view, proj, planes = S.make_camera(W, H, 1200.0, 1200.0, (0.4, -0.1, 0.3))
    ref = oracle.render_forward(scene, view, proj, planes, H, W, 3)
    alloc = ref.alloc[0]
    assert (alloc > 1024).sum() >= 20 and (alloc > 4096).sum() >= 3, ((alloc > 1024).sum(), (alloc > 4096).sum(), int(alloc.max()))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    op = ref.act[4]
    ks, vs = F.create_table(dev(ref.ndc), dev(ref.inv_cov), dev(op), dev(ref.prefix), dev(ref.depth_sorted_index), None, None, H, W, 8, 16)
    assert np.array_equal(ks.cpu().numpy(), ref.sorted_tile) and np.array_equal(vs.cpu().numpy(), ref.sorted_point)
    total = int(ref.prefix[0, -1])
    want = int(1.5 * int(0.45 * total))                         # sized from a much smaller "previous frame": drops whole splats
    ks_r, vs_r, _, _ = oracle.create_table(ref.ndc, ref.inv_cov, op, ref.prefix, ref.depth_sorted_index, H, W, 8, 16, table_len=want)
    fb = torch.tensor([int(0.45 * total)], dtype=torch.int32).pin_memory()
    ks, vs = F.create_table(dev(ref.ndc), dev(ref.inv_cov), dev(op), dev(ref.prefix), dev(ref.depth_sorted_index), fb, torch.tensor([0]),
                            H, W, 8, 16)
    torch.cuda.synchronize()
    assert want < total and np.array_equal(ks.cpu().numpy(), ks_r)
    live = ks_r[0] > 0
    assert np.array_equal(vs.cpu().numpy()[0][live], vs_r[0][live])


# Needle-like conics (b^2 ~ a c) and splats at the 1/255 opacity cut-off whose FIRST tile slice has both lines unselected and neither
# extreme point inside: max_tile_v < min_tile_v, and the reference adds that NEGATIVE difference to the splat's tile count
# (GR/speedy_splat.cuh:118-125) while emitting nothing for the slice.  Found by tests/host/walk_check.cpp's generator (1920x1080, 8x16);
# (ndc.x, ndc.y, conic a, b, c, opacity).
DEGENERATE_SPLATS = [
    (-3.866354823e-01, 4.910708591e-02, 1.318661403e-02, 1.318791416e-02, 1.318921614e-02, 3.921945114e-03),   # 54 tiles counted, 63 slices, slice 0: 122..19
    (-4.079762995e-01, -1.158427820e-01, 6.700431928e-03, 2.074851654e-02, 6.424973905e-02, 3.921761177e-03),   # 44 tiles counted, 57 slices, slice 0: 72..0
    (8.839730918e-02, -9.655379690e-03, 1.324557648e+02, 1.950504456e+02, 2.872255554e+02, 1.000000000e+00),   # 103 tiles counted, 120 slices, slice 0: 135..0
    (-3.678972721e-01, 8.517017215e-02, 1.051743701e-02, 1.052330341e-02, 1.052917447e-02, 3.921788651e-03),   # 38 tiles counted, 56 slices, slice 0: 108..38
    (1.114281714e-01, 7.290778756e-01, 5.003720398e+02, 5.004038391e+02, 5.004356689e+02, 1.000000000e+00),   # 46 tiles counted, 104 slices, slice 0: 135..0
    (-2.951445282e-01, -1.807289720e-01, 1.067461446e-02, 1.067493390e-02, 1.067525428e-02, 3.921690397e-03),   # 40 tiles counted, 62 slices, slice 0: 92..18
    (-5.725104809e-01, 7.241415381e-01, 1.948983409e-02, 4.034486786e-02, 8.351577073e-02, 3.921918105e-03),   # 33 tiles counted, 42 slices, slice 0: 131..0
    (-1.365235746e-01, 4.870536625e-01, 4.218530841e-03, 4.218396265e-03, 4.218262620e-03, 3.921733238e-03),   # 38 tiles counted, 72 slices, slice 0: 135..62
    (7.500848174e-01, -4.425054789e-01, 3.454413672e-04, 3.140928748e-04, 2.855892526e-04, 3.921702038e-03),   # 55 tiles counted, 120 slices, slice 0: 135..0
    (1.656607985e-01, -1.255498528e-01, 7.024222054e-03, 5.824854132e-03, 4.830275662e-03, 3.921741154e-03),   # 45 tiles counted, 92 slices, slice 0: 110..8
    (-2.506476343e-01, 4.049793780e-01, 3.102688119e-03, 3.377848538e-03, 3.677412868e-03, 3.921753261e-03),   # 43 tiles counted, 65 slices, slice 0: 130..60
    (-9.531433135e-02, -4.155531228e-01, 3.043316538e-04, 3.043696343e-04, 3.044076730e-04, 3.921619616e-03),   # 67 tiles counted, 101 slices, slice 0: 132..0
    (2.229391038e-01, -4.387749732e-01, 3.448471427e-03, -7.875907235e-03, 1.798765734e-02, 3.921882715e-03),   # 74 tiles counted, 73 slices, slice 0: 112..35
    (-1.885991544e-01, 4.217935205e-01, 4.597336520e-03, 7.089260500e-03, 1.093189977e-02, 3.921925556e-03),   # 66 tiles counted, 82 slices, slice 0: 135..54
]


def degenerate_table_inputs(copies=40, seed=3):
    """(ndc [1,4,N], inv_cov [1,2,2,N], opacity [1,N], view depth [1,N]): `copies` of every degenerate splat, shuffled.  256 consecutive
    slots then hold ~2x the key emission's in-workgroup budget (8192 staged keys), so its small / big threshold drops to 32 tiles and every
    one of these splats takes the cooperative big-splat path (csrc/binning.hip dup_big_kernel), where a negative slice count used to put a
    slice start at a negative output position."""
    rng = np.random.default_rng(seed)
    rows = np.array(DEGENERATE_SPLATS, dtype=np.float32)
    idx = rng.permutation(np.repeat(np.arange(len(rows)), copies))
    r = rows[idx]
    N = len(r)
    ndc = np.zeros((1, 4, N), np.float32); ndc[0, 0] = r[:, 0]; ndc[0, 1] = r[:, 1]; ndc[0, 2] = 0.5; ndc[0, 3] = 1.0
    inv = np.zeros((1, 2, 2, N), np.float32); inv[0, 0, 0] = r[:, 2]; inv[0, 0, 1] = inv[0, 1, 0] = r[:, 3]; inv[0, 1, 1] = r[:, 4]
    op = np.ascontiguousarray(r[:, 5][None])
    vz = (1.0 + rng.random((1, N))).astype(np.float32)
    return ndc, inv, op, vz


def test_degenerate_slices_keep_the_table_inside_each_splats_share(oracle):
    """a splat whose tile count contains a negative slice owns exactly its share of the table: the first `count` tiles in slice order, every
    key a valid tile id, in the small-splat and in the big-splat path of the key emission; bit-exact against the oracle (whose restatement
    bounds the emission the same way, oracle/litegs_oracle.c process_tiles)"""
    from litegs_amd import fused as F
    from litegs_amd.fast import FusedRenderer
    H, W = 1080, 1920
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    FusedRenderer.sanitised_counts(reset=True)
    for copies in (1, 40):                                   # 1: every splat alone in its group (in-workgroup path); 40: the big-splat path
        ndc, inv, op, vz = degenerate_table_inputs(copies)
        N = ndc.shape[-1]
        _, _, al_r = oracle.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
        assert (al_r[0] > 32).all(), al_r[0].min()
        _, _, al = F.get_allocate_size(dev(ndc), dev(vz), dev(inv), dev(op), H, W, 8, 16, None)
        assert np.array_equal(al.cpu().numpy(), al_r)
        dsi = np.argsort(vz, axis=-1, kind="stable").astype(np.int64)
        prefix = np.cumsum(np.take_along_axis(al_r, dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
        ks_r, vs_r, _, _ = oracle.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
        assert (ks_r > 0).all() and ks_r.max() <= 135 * 120     # the oracle fills every share with valid keys
        ks, vs = F.create_table(dev(ndc), dev(inv), dev(op), dev(prefix), dev(dsi), None, None, H, W, 8, 16)
        torch.cuda.synchronize()
        assert np.array_equal(ks.cpu().numpy(), ks_r), copies
        assert np.array_equal(vs.cpu().numpy(), vs_r), copies
    counts = FusedRenderer.sanitised_counts(reset=True)
    assert not any(v for k, v in counts.items() if k != "truncated_tables"), counts     # nothing had to be neutralised


EMISSION_VARIANTS = {                       # lg_set_tuning(10, ceiling): largest tile count the owning thread of dup_small walks itself (csrc/binning.hip)
    "ceiling_256": 256,                     # the default
    "ceiling_32": 32,                       # everything beyond 32 tiles goes to the cooperative kernel (dup_big)
    "ceiling_128": 128,
}


@pytest.mark.parametrize("variant", list(EMISSION_VARIANTS))
def test_emission_variants_build_one_table(oracle, variant):
    """the two size classes of the key emission (walked by the owning thread into the workgroup's LDS buffer / queued for the cooperative
    kernel, one lane per slice) are the same walk: wherever the ceiling between them sits, create_table returns the oracle's table bit for
    bit -- on the degenerate splats (negative first slice), on near-camera giants (1 000..16 000 tiles) and on an exact and a truncated table"""
    from litegs_amd import fused as F
    from litegs_amd._lib import lib, check
    from litegs_amd.fast import FusedRenderer
    ceiling = EMISSION_VARIANTS[variant]
    H, W = 1080, 1920
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    FusedRenderer.sanitised_counts(reset=True)
    check(lib().lg_set_tuning(10, ceiling), "tuning")
    try:
        for copies in (1, 40):
            ndc, inv, op, vz = degenerate_table_inputs(copies)
            _, _, al_r = oracle.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
            dsi = np.argsort(vz, axis=-1, kind="stable").astype(np.int64)
            prefix = np.cumsum(np.take_along_axis(al_r, dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
            ks_r, vs_r, _, _ = oracle.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
            ks, vs = F.create_table(dev(ndc), dev(inv), dev(op), dev(prefix), dev(dsi), None, None, H, W, 8, 16)
            torch.cuda.synchronize()
            assert np.array_equal(ks.cpu().numpy(), ks_r) and np.array_equal(vs.cpu().numpy(), vs_r), copies
        scene = list(S.make_scene(2048, seed=21, scale_mult=1.5))
        view, proj, planes = S.make_camera(W, H, 1200.0, 1200.0, (0.4, -0.1, 0.3))
        ref = oracle.render_forward(scene, view, proj, planes, H, W, 3)
        assert (ref.alloc[0] > 4096).sum() >= 3
        op = ref.act[4]
        ks, vs = F.create_table(dev(ref.ndc), dev(ref.inv_cov), dev(op), dev(ref.prefix), dev(ref.depth_sorted_index), None, None, H, W, 8, 16)
        assert np.array_equal(ks.cpu().numpy(), ref.sorted_tile) and np.array_equal(vs.cpu().numpy(), ref.sorted_point)
        total = int(ref.prefix[0, -1])
        want = int(1.5 * int(0.45 * total))
        ks_r, vs_r, _, _ = oracle.create_table(ref.ndc, ref.inv_cov, op, ref.prefix, ref.depth_sorted_index, H, W, 8, 16, table_len=want)
        fb = torch.tensor([int(0.45 * total)], dtype=torch.int32).pin_memory()
        ks, vs = F.create_table(dev(ref.ndc), dev(ref.inv_cov), dev(op), dev(ref.prefix), dev(ref.depth_sorted_index), fb, torch.tensor([0]), H, W, 8, 16)
        torch.cuda.synchronize()
        live = ks_r[0] > 0
        assert want < total and np.array_equal(ks.cpu().numpy(), ks_r) and np.array_equal(vs.cpu().numpy()[0][live], vs_r[0][live])
    finally:
        check(lib().lg_set_tuning(10, 256), "tuning")
    counts = FusedRenderer.sanitised_counts(reset=True)
    assert not any(v for k, v in counts.items() if k != "truncated_tables"), counts
