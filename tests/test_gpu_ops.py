"""GPU parity tests, op by op: every entry point of the C ABI (called through the ``litegs_fused`` surface)
against the CPU oracle on identical seeded inputs.  Integer/index outputs are compared bit-exactly; float
outputs to 1e-5 (these ops have no decision thresholds)."""
import numpy as np
import pytest
import torch

from tests.util import assert_bracket, assert_close, bracket_of, bracket_variants, case, oracle_forward, d_img_for, GRAD_ROUND_ATOL, GRAD_ROUND_MAX

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", params=["ctypes", "ext"])
def F(request):
    """the litegs_fused surface through both bindings of the C ABI: ctypes (litegs_amd/fused.py) and the compiled torch extension"""
    from litegs_amd import binding, fused
    if request.param == "ctypes":
        return fused
    if binding.compiled is None:
        pytest.skip("compiled litegs_fused extension not built")
    return binding.ops


@pytest.fixture(params=["lds", "ballot"])
def rank_mode(request):
    """both rankings of the radix passes: lane-ordered returning LDS adds (self-tested fast path) and the ballot ranking"""
    from litegs_amd._lib import lib
    L = lib()
    L.lg_radix_set_rank_mode(0 if request.param == "lds" else 1)
    yield request.param
    L.lg_radix_set_rank_mode(-1)


def test_radix_rank_selftest_runs():
    from litegs_amd._lib import lib
    L = lib()
    L.lg_radix_set_rank_mode(-1)
    mode = L.lg_radix_rank_mode()
    assert mode in (0, 1)
    print(f"[radix] rank self-test: {'lane-ordered LDS returns verified' if mode == 0 else 'VIOLATED -> ballot ranking'}")


def test_frustum_culling_bit_exact(F, oracle):
    c = case("small")
    origin, ext = oracle.cluster_AABB(*c["params"][:3])
    vis_ref, ids_ref = oracle.frustum_culling_aabb(origin, ext, c["planes"])
    vis, num, ids = F.frustum_culling_aabb(dev(origin), dev(ext), dev(c["planes"]), None, None)
    assert int(num.item()) == len(ids_ref)
    assert np.array_equal(host(vis), vis_ref)
    assert np.array_equal(host(ids), ids_ref)
    assert 0 < len(ids_ref) < origin.shape[1], "case must cull some but not all chunks"


def test_frustum_culling_feedback_protocol(F, oracle):
    c = case("small")
    origin, ext = oracle.cluster_AABB(*c["params"][:3])
    _, ids_ref = oracle.frustum_culling_aabb(origin, ext, c["planes"])
    fb = torch.zeros((4,), dtype=torch.int32).pin_memory()
    idx = torch.tensor([2], dtype=torch.int64)
    _, num, ids = F.frustum_culling_aabb(dev(origin), dev(ext), dev(c["planes"]), fb, idx)   # first time: blocking, exact
    torch.cuda.synchronize()
    assert ids.shape[0] == len(ids_ref) and int(fb[2]) == len(ids_ref)
    _, num, ids = F.frustum_culling_aabb(dev(origin), dev(ext), dev(c["planes"]), fb, idx)   # second time: 1.2x prediction
    assert ids.shape[0] == min(int(1.2 * len(ids_ref)), origin.shape[1])
    assert np.array_equal(host(ids)[:len(ids_ref)], ids_ref)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_activate_forward_backward(F, oracle, degree):
    c = case("small")
    res = oracle_forward("small")
    params = c["params"]
    nvis = res.nvis
    A = nvis + 3                                             # over-allocated, as with the 1.2x prediction
    ids = np.concatenate([res.visible_chunkid, np.arange(3)]).astype(np.int64)
    ref = oracle.activate_forward(degree, ids, nvis, c["view"], *params, alloc=A)
    num = torch.tensor([nvis], dtype=torch.int32).cuda()
    out = F.cull_compact_activate(degree, dev(ids), num, dev(c["view"]), *[dev(p) for p in params])
    for o, r, n in zip(out, ref, ["pos", "scale", "rot", "color", "opacity"]):
        assert_close(host(o)[..., :nvis, :], r[..., :nvis, :], atol=2e-6, normalize=True, name=f"activate.{n}")
    assert (host(out[4])[:, nvis:, :] == 0).all(), "tail chunks must have zero opacity"

    rng = np.random.default_rng(3)
    g = [rng.standard_normal(r.shape).astype(np.float32) for r in ref]
    dref = oracle.activate_backward(degree, ids, nvis, c["view"], *params, *g)
    dout = F.activate_backward(degree, dev(ids), num, dev(c["view"]), *[dev(p) for p in params], *[dev(x) for x in g])
    for o, r, n in zip(dout, dref, ["xyz", "scale", "rot", "sh0", "shr", "opacity"]):
        assert_close(host(o)[..., :nvis, :], r[..., :nvis, :], atol=2e-6, normalize=True, name=f"activate_backward.{n}")
    assert (host(dout[4]) [(degree + 1) ** 2 - 1:] == 0).all(), "inactive SH degrees must get zero gradient"


def test_projection_chain_forward_backward(F, oracle):
    c = case("small")
    res = oracle_forward("small")
    pos, sc, rt, col, op = res.act
    N = pos.shape[1]
    valid = N - 100
    vl = torch.tensor([valid], dtype=torch.int32).cuda()
    view, proj = c["view"], c["proj"]
    H, W = c["H"], c["W"]

    vp, ndc = F.mvp_transform_forward(dev(pos), dev(view), dev(proj), vl)
    vp_r, ndc_r = oracle.mvp_forward(pos, view, proj, valid)
    assert_close(host(vp)[..., :valid], vp_r[..., :valid], atol=1e-6, normalize=True, name="view_pos")
    assert_close(host(ndc)[..., :valid], ndc_r[..., :valid], atol=1e-6, normalize=True, name="ndc")

    T = F.createTransformMatrix_forward(dev(rt), dev(sc), vl)
    T_r = oracle.transform_matrix_forward(rt, sc, valid)
    assert_close(host(T)[..., :valid], T_r[..., :valid], atol=1e-6, normalize=True, name="T")

    J = F.jacobianRayspace(dev(vp_r), dev(proj), H, W, vl)
    J_r = oracle.jacobian_rayspace(vp_r, proj, H, W, valid)
    assert_close(host(J), J_r, atol=1e-6, normalize=True, name="J")

    cov = F.createCov2dDirectly_forward(dev(J_r), dev(view), dev(T_r), vl)
    cov_r = oracle.cov2d_forward(J_r, view, T_r, valid)
    assert_close(host(cov)[..., :valid], cov_r[..., :valid], atol=1e-6, normalize=True, name="cov2d")

    val, vec, inv = F.eigh_and_inv_2x2matrix_forward(dev(cov_r), vl)
    val_r, vec_r, inv_r = oracle.eigh_inv_forward(cov_r, valid)
    assert_close(host(val)[..., :valid], val_r[..., :valid], atol=1e-6, normalize=True, name="eig_val")
    assert_close(host(vec)[..., :valid], vec_r[..., :valid], atol=1e-5, name="eig_vec")
    assert_close(host(inv)[..., :valid], inv_r[..., :valid], atol=1e-6, normalize=True, name="inv_cov")

    rng = np.random.default_rng(5)
    g_inv = rng.standard_normal(inv_r.shape).astype(np.float32)
    g_inv[:, 0, 1] = g_inv[:, 1, 0]
    g_cov = F.inv_2x2matrix_backward(dev(inv_r), dev(g_inv), vl)
    g_cov_r = oracle.inv2x2_backward(inv_r, g_inv, valid)
    assert_close(host(g_cov)[..., :valid], g_cov_r[..., :valid], atol=1e-6, normalize=True, name="g_cov2d")

    gT = F.createCov2dDirectly_backward(dev(g_cov_r), dev(J_r), dev(view), dev(T_r), vl)
    gT_r = oracle.cov2d_backward(g_cov_r, J_r, view, T_r, valid)
    assert_close(host(gT), gT_r, atol=1e-6, normalize=True, name="gT")

    gq, gs = F.createTransformMatrix_backward(dev(gT_r), dev(rt), dev(sc), vl)
    gq_r, gs_r = oracle.transform_matrix_backward(gT_r, rt, sc, valid)
    assert_close(host(gq)[..., :valid], gq_r[..., :valid], atol=1e-6, normalize=True, name="g_quat")
    assert_close(host(gs)[..., :valid], gs_r[..., :valid], atol=1e-6, normalize=True, name="g_scale")

    g_ndc = rng.standard_normal(ndc_r.shape).astype(np.float32)
    g_view = rng.standard_normal(vp_r.shape).astype(np.float32)
    gw = F.mvp_transform_backward(dev(g_ndc), dev(g_view), dev(view), dev(proj), dev(vp_r), vl)
    gw_r = oracle.mvp_backward(g_ndc, g_view, view, proj, vp_r, valid)
    assert_close(host(gw)[..., :valid], gw_r[..., :valid], atol=1e-6, normalize=True, name="g_world")


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh2rgb(F, oracle, degree):
    rng = np.random.default_rng(7)
    N, V, R = 5000, 2, 15
    sh0 = rng.standard_normal((1, 3, N)).astype(np.float32)
    shr = rng.standard_normal((R, 3, N)).astype(np.float32)
    d = rng.standard_normal((V, 3, N)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rgb = F.sh2rgb_forward(degree, dev(sh0), dev(shr), dev(d))
    assert_close(host(rgb), oracle.sh2rgb_forward(degree, sh0, shr, d), atol=2e-6, normalize=True, name="sh2rgb")
    g = rng.standard_normal((V, 3, N)).astype(np.float32)
    d0, dr, dd = F.sh2rgb_backward(degree, dev(g), R, dev(d), dev(sh0), dev(shr))
    r0, rr, rd = oracle.sh2rgb_backward(degree, g, R, d)
    assert_close(host(d0), r0, atol=2e-6, normalize=True, name="d_sh0")
    assert_close(host(dr), rr, atol=2e-6, normalize=True, name="d_shr")
    assert (host(dd) == 0).all()


@pytest.mark.parametrize("name", ["small", "pad"])
def test_binning_bit_exact(F, oracle, name):
    """get_allocate_size -> (stable depth order) -> create_table -> tileRange: integer outputs, bit-exact."""
    c = case(name)
    res = oracle_forward(name)
    H, W = c["H"], c["W"]
    pos, sc, rt, col, op = res.act
    N = pos.shape[1]
    vd = np.ascontiguousarray(res.view_pos[:, 2, :])
    lu, rd, al = F.get_allocate_size(dev(res.ndc), dev(vd), dev(res.inv_cov), dev(op), H, W, 8, 16, None)
    lu_r, rd_r, al_r = oracle.get_allocate_size(res.ndc, vd, res.inv_cov, op, H, W, 8, 16)
    assert np.array_equal(host(al), al_r)
    vis = al_r[0] > 0
    assert vis.sum() > 100
    assert np.array_equal(host(lu)[:, :, vis], lu_r[:, :, vis]) and np.array_equal(host(rd)[:, :, vis], rd_r[:, :, vis])

    ks, vs = F.create_table(dev(res.ndc), dev(res.inv_cov), dev(op), dev(res.prefix), dev(res.depth_sorted_index), None, None, H, W, 8, 16)
    assert np.array_equal(host(ks), res.sorted_tile)
    assert np.array_equal(host(vs), res.sorted_point)
    ntiles = ((H + 7) // 8) * ((W + 15) // 16)
    tr = F.tileRange(ks, ntiles)
    assert np.array_equal(host(tr), res.tile_start)


@pytest.mark.parametrize("name", ["small", "pad"])
def test_grouped_binning_returns_the_sorted_pipelines_table(oracle, name):
    """litegs_amd.binning.binning (the host mirror of Binning.call_fused, wrapper.py:717-763) builds its table without the two full-length
    sorts (emission in id order, grouping by tile, per-tile depth sort): tile ranges and point ids must be bit for bit what the reference's
    structure (stable depth sort, create_table, tileRange) yields -- exact first visit and 1.5x over-allocated revisit"""
    from litegs_amd import binning as B
    c = case(name)
    res = oracle_forward(name)
    H, W = c["H"], c["W"]
    op = res.act[4]
    vd = np.ascontiguousarray(res.view_pos[:, 2, :])
    total = int(res.prefix[0, -1])
    ntiles = ((H + 7) // 8) * ((W + 15) // 16)
    fb = torch.zeros((1,), dtype=torch.int32).pin_memory()
    idx = torch.tensor([0])
    for visit in range(2):                       # first visit: blocking exact size; second: 1.5 x the total the first one fed back
        want = total if visit == 0 else int(1.5 * total)
        ks_r, vs_r, _, _ = oracle.create_table(res.ndc, res.inv_cov, op, res.prefix, res.depth_sorted_index, H, W, 8, 16, table_len=want)
        assert B._GROUPED
        ts, sp, nvis = B.binning(dev(res.ndc), dev(vd), dev(res.inv_cov), dev(op), None, fb, idx, (H, W), (8, 16))
        torch.cuda.synchronize()
        assert sp.shape[1] == want and int(fb[0]) == total
        live = ks_r[0] > 0                        # payload of the zero padding is undefined
        assert np.array_equal(host(sp)[0][live], vs_r[0][live])
        B._GROUPED = False
        try:
            fb2 = fb.clone().pin_memory()
            if visit == 0:
                fb2.zero_()
            ts2, sp2, nvis2 = B.binning(dev(res.ndc), dev(vd), dev(res.inv_cov), dev(op), None, fb2, idx, (H, W), (8, 16))
        finally:
            B._GROUPED = True
        assert np.array_equal(host(ts), host(ts2)) and np.array_equal(host(nvis), host(nvis2))
        assert np.array_equal(host(sp2)[0][live], vs_r[0][live])
        if visit == 0:
            assert np.array_equal(host(ts), res.tile_start)


def test_create_table_overallocated_and_truncated(F, oracle):
    """GPU-driven sizing: a 1.5x over-allocated table sorts its zero padding to the front (tile 0); an under-sized
    table silently drops whole splats (GR/binning.cu:63)."""
    c = case("small")
    res = oracle_forward("small")
    H, W = c["H"], c["W"]
    op = res.act[4]
    total = int(res.prefix[0, -1])
    for last_epoch_total in (total, int(0.4 * total)):
        want = int(1.5 * last_epoch_total)
        ks_r, vs_r, _, _ = oracle.create_table(res.ndc, res.inv_cov, op, res.prefix, res.depth_sorted_index, H, W, 8, 16, table_len=want)
        fb = torch.tensor([last_epoch_total], dtype=torch.int32).pin_memory()
        ks, vs = F.create_table(dev(res.ndc), dev(res.inv_cov), dev(op), dev(res.prefix), dev(res.depth_sorted_index), fb,
                                torch.tensor([0]), H, W, 8, 16)
        torch.cuda.synchronize()
        assert ks.shape[1] == want and int(fb[0]) == total, "feedback buffer must receive this frame's exact total"
        assert np.array_equal(host(ks), ks_r)
        live = ks_r[0] > 0                      # payload of the zero padding is undefined
        assert np.array_equal(host(vs)[0][live], vs_r[0][live])
        assert (live.sum() < total) == (want < total)


@pytest.mark.parametrize("n,bits", [(1, 8), (255, 8), (4096, 8), (4097, 14), (100_003, 14), (1_000_000, 11), (300_000, 32)])
def test_radix_sort_is_stable_and_exact(F, oracle, rank_mode, n, bits):
    rng = np.random.default_rng(n)
    hi = (1 << bits) - 1 if bits < 32 else 0xFFFFFFFF
    keys = rng.integers(0, hi, size=n, dtype=np.uint64, endpoint=True).astype(np.uint32)
    if n > 1000:
        keys[::7] = keys[3]                     # many duplicates: stability is observable through the payload
    vals = np.arange(n, dtype=np.uint32)
    k, v = F.radix_sort_pairs(dev(keys.view(np.int32)), dev(vals.view(np.int32)), 0, bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(host(k).view(np.uint32), keys[order])
    assert np.array_equal(host(v).view(np.uint32), vals[order])


@pytest.mark.parametrize("pattern", ["all_equal", "two_values", "three_bit", "runs_of_64", "tile_like"])
def test_radix_sort_stability_under_heavy_duplicates(F, rank_mode, pattern):
    """Equal digits inside one 64-key wave instruction must keep their input order (the ranking relies on the LDS serving
    same-address returning adds in lane order): patterns where every instruction has many collisions, 1.2 M keys so that the
    look-back chain spans ~300 workgroups."""
    n = 1_200_011
    rng = np.random.default_rng(7)
    if pattern == "all_equal":
        keys = np.full(n, 0x1234, np.uint32)
    elif pattern == "two_values":
        keys = np.where(rng.random(n) < 0.5, 0x00ff, 0x1f00).astype(np.uint32)
    elif pattern == "three_bit":
        keys = (rng.integers(0, 8, n) * 0x0421).astype(np.uint32)
    elif pattern == "runs_of_64":
        keys = (np.arange(n) // 64 % 5000).astype(np.uint32)
    else:                                   # neighbouring tile ids, as duplicate_with_keys emits them
        base = rng.integers(1, 8000, n // 16 + 1)
        keys = (base[:, None] + (np.arange(16) % 4)[None, :] + 120 * (np.arange(16) // 4)[None, :]).reshape(-1)[:n].astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k, v = F.radix_sort_pairs(dev(keys.view(np.int32)), dev(vals.view(np.int32)), 0, 14)
    order = np.argsort(keys & 0x3fff, kind="stable")
    assert np.array_equal(host(k).view(np.uint32), keys[order])
    assert np.array_equal(host(v).view(np.uint32), vals[order])


def test_depth_sort_and_scan_match_numpy(F):
    from litegs_amd import binning
    rng = np.random.default_rng(11)
    n = 70_001
    depth = rng.standard_normal(n).astype(np.float32) * 10
    depth[::5] = depth[1]
    alloc = rng.integers(0, 40, size=n).astype(np.int32)
    idx, prefix = binning.depth_order_and_prefix(dev(depth[None]), dev(alloc[None]))
    order = np.argsort(depth, kind="stable")
    assert np.array_equal(host(idx)[0], order)
    assert np.array_equal(host(prefix)[0], np.cumsum(alloc[order]).astype(np.int32))


@pytest.mark.parametrize("name", ["small", "pad"])
def test_raster_forward(F, oracle, name):
    c = case(name)
    res = oracle_forward(name, stat=True)
    H, W = c["H"], c["W"]
    pos, sc, rt, col, op = res.act
    out = F.rasterize_forward(dev(res.sorted_point), dev(res.tile_start), dev(res.ndc), dev(res.inv_cov), dev(col), dev(op), None,
                              H, W, 8, 16, True, False, False)
    img, trans, depth, last, packed, fc, fw = out
    (_, var) = bracket_of(oracle, lambda: oracle.raster_forward(res.sorted_point, res.tile_start, res.packed, H, W, 8, 16, enable_stat=True))
    assert_bracket(host(img), res.img, [v[0] for v in var], name="img", decided_max=20)
    assert_bracket(host(trans), res.trans, [v[1] for v in var], name="transmitance", decided_max=20)
    lastd = np.abs(host(last).astype(np.int32) - res.last.astype(np.int32))
    assert (lastd > 0).mean() <= 2e-5 and lastd.max() <= 2, "last_contributor"
    fcd = np.abs(host(fc).astype(np.int64) - res.frag_count)
    assert fcd.max() <= 2 and (fcd > 0).mean() < 1e-3, "fragment_count"
    assert_bracket(host(fw), res.frag_weight, [v[4] for v in var], atol=1e-4, normalize=True, name="fragment_weight_sum", decided_max=60)
    assert res.img.max() > 0.2


@pytest.mark.parametrize("name,trans", [("small", False), ("pad", False), ("small", True)])
def test_raster_backward(F, oracle, name, trans):
    c = case(name)
    res = oracle_forward(name)
    H, W = c["H"], c["W"]
    col, op = res.act[3], res.act[4]
    d_img = d_img_for(res)
    d_trans = d_img_for(res, 9)[:, :1].copy() if trans else None
    ref, var = bracket_of(oracle, lambda: oracle.raster_backward(res.sorted_point, res.tile_start, res.packed, res.trans, res.last, d_img, H, W, 8, 16,
                                                                 d_trans=d_trans, inv_scaler=0.5, enable_stat=True))
    fwd = F.rasterize_forward(dev(res.sorted_point), dev(res.tile_start), dev(res.ndc), dev(res.inv_cov), dev(col), dev(op), None,
                              H, W, 8, 16, False, trans, False)
    packed = fwd[4]
    got = F.rasterize_backward(dev(res.sorted_point), dev(res.tile_start), packed, None, dev(res.trans), dev(res.last), dev(d_img),
                               dev(d_trans) if trans else None, None, torch.tensor(0.5).cuda(), H, W, 8, 16, True)
    names = ["d_ndc", "d_cov2d_inv", "d_color", "d_opacity", "err_sum", "err_square_sum"]
    for k, (g, r, n) in enumerate(zip([got[0], got[1], got[2], got[3], got[5]], ref, [names[0], names[1], names[2], names[3], names[5]])):
        assert_bracket(host(g), r, [v[k] for v in var], atol=1e-4, normalize=True, name=n, decided_max=40, round_atol=GRAD_ROUND_ATOL, round_max=GRAD_ROUND_MAX)
    assert (host(got[4]) == 0).all()
    assert np.abs(ref[0]).max() > 0


@pytest.mark.parametrize("name", ["small", "pad"])
def test_raster_backward_splat_parallel_variant(F, oracle, name):
    """the splat-parallel formulation of the blend backward (csrc/raster.hip raster_backward_sp_kernel, lg_set_tuning(5, 2): an A/B variant,
    not the default) against the oracle and against the default kernel"""
    from litegs_amd._lib import check, lib
    c = case(name)
    res = oracle_forward(name)
    H, W = c["H"], c["W"]
    col, op = res.act[3], res.act[4]
    d_img = d_img_for(res)
    ref, var = bracket_of(oracle, lambda: oracle.raster_backward(res.sorted_point, res.tile_start, res.packed, res.trans, res.last, d_img, H, W, 8, 16, inv_scaler=0.5))
    fwd = F.rasterize_forward(dev(res.sorted_point), dev(res.tile_start), dev(res.ndc), dev(res.inv_cov), dev(col), dev(op), None,
                              H, W, 8, 16, False, False, False)
    packed = fwd[4]
    out = {}
    try:
        for variant in (2, 1):
            check(lib().lg_set_tuning(5, variant), "lg_set_tuning")
            out[variant] = F.rasterize_backward(dev(res.sorted_point), dev(res.tile_start), packed, None, dev(res.trans), dev(res.last), dev(d_img),
                                                None, None, torch.tensor(0.5).cuda(), H, W, 8, 16, False)
            torch.cuda.synchronize()
    finally:
        check(lib().lg_set_tuning(5, 1), "lg_set_tuning")
    for k, (g, g1, r, n) in enumerate(zip(out[2][:4], out[1][:4], ref[:4], ["d_ndc", "d_cov2d_inv", "d_color", "d_opacity"])):
        assert_bracket(host(g), r, [v[k] for v in var], atol=1e-4, normalize=True, name=n + " (splat-parallel)", decided_max=40, round_atol=GRAD_ROUND_ATOL, round_max=GRAD_ROUND_MAX)
        scale = max(float(np.abs(r).max()), 1e-30)
        assert np.abs(host(g) - host(g1)).max() / scale < 2e-4, n           # the two kernels: rounding and the rare threshold flip only


def test_raster_specific_tiles(F, oracle):
    """specific_tiles = heavy-first schedule (statistic_helper.py:77): same image, any order; 0 entries are skipped."""
    c = case("small")
    res = oracle_forward("small")
    H, W = c["H"], c["W"]
    col, op = res.act[3], res.act[4]
    ntiles = ((H + 7) // 8) * ((W + 15) // 16)
    rng = np.random.default_rng(2)
    order = (rng.permutation(ntiles) + 1).astype(np.int32)
    order = np.concatenate([order, np.zeros(3, np.int32)])[None]
    out = F.rasterize_forward(dev(res.sorted_point), dev(res.tile_start), dev(res.ndc), dev(res.inv_cov), dev(col), dev(op), dev(order),
                              H, W, 8, 16, False, False, False)
    assert_bracket(host(out[0]), res.img, [v.img for v, _ in bracket_variants(oracle, res, H, W)], name="img(specific_tiles)", decided_max=20)


def test_adam_and_sparse_scatter(F, oracle):
    rng = np.random.default_rng(13)
    E, chunks, S, A, nvis = 45, 50, 128, 30, 23
    p = rng.standard_normal((E, chunks, S)).astype(np.float32)
    m = rng.standard_normal((E, chunks, S)).astype(np.float32) * 0.1
    v = rng.random((E, chunks, S)).astype(np.float32) * 0.01
    g = rng.standard_normal((E, A, S)).astype(np.float32)
    ids = rng.permutation(chunks)[:A].astype(np.int64)
    pd, md, vd = dev(p), dev(m), dev(v)
    F.adamUpdate(pd, dev(g), md, vd, dev(ids), torch.tensor([nvis], dtype=torch.int32).cuda(), 1e-2, 0.9, 0.999, 1e-15)
    oracle.adam_chunk(p, g, m, v, ids, nvis, 1e-2)
    assert_close(host(pd), p, atol=1e-6, normalize=True, name="adam.param")
    assert_close(host(md), m, atol=1e-6, normalize=True, name="adam.m")
    assert_close(host(vd), v, atol=1e-6, normalize=True, name="adam.v")

    N = 3000
    p2 = rng.standard_normal((3, N)).astype(np.float32); m2 = np.zeros_like(p2); v2 = np.zeros_like(p2)
    g2 = rng.standard_normal((3, N)).astype(np.float32)
    mask = (rng.random(N) > 0.5).astype(np.int64)
    p2d, m2d, v2d = dev(p2), dev(m2), dev(v2)
    F.adamUpdate(p2d, dev(g2), m2d, v2d, dev(mask), None, 1e-3, 0.9, 0.999, 1e-15)
    oracle.adam_primitive(p2, g2, m2, v2, mask, 1e-3)
    assert_close(host(p2d), p2, atol=1e-6, normalize=True, name="adam2.param")

    for dt, op in ((np.float32, "add"), (np.int32, "add"), (np.float32, "max"), (np.int32, "min")):
        Aa = (rng.standard_normal((2, chunks, S)) * 10).astype(dt)
        Bb = (rng.standard_normal((2, A, S)) * 10).astype(dt)
        Ad = dev(Aa)
        F.gpu_driven_pipeline_sparse_op(Ad, dev(Bb), dev(ids), torch.tensor([nvis], dtype=torch.int32).cuda(), op)
        oracle.sparse_scatter(Aa, Bb, ids, nvis, op)
        assert np.array_equal(host(Ad), Aa), f"sparse {dt} {op}"


def test_cpu_tensors_are_rejected(F):
    with pytest.raises(RuntimeError):
        F.createTransformMatrix_forward(torch.zeros(4, 8), torch.zeros(3, 8), None)


def test_radix_sort_device_bounded_prefix(F):
    """lg_radix_sort_pairs_bounded / lg_tile_range_bounded: only the first *n_dev entries form the table (GPU-driven sizing)."""
    from litegs_amd._lib import lib, check
    L = lib()
    rng = np.random.default_rng(5)
    n, live, bits, ntiles = 50_000, 33_333, 14, 16200
    keys = rng.integers(1, ntiles + 1, size=n).astype(np.int32)
    vals = np.arange(n, dtype=np.int32)
    ka, va = dev(keys), dev(vals)
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    n_dev = torch.tensor([live], dtype=torch.int32).cuda()
    tb = L.lg_radix_sort_temp_bytes(n)
    temp = torch.empty((tb,), dtype=torch.uint8).cuda()
    s = torch.cuda.current_stream().cuda_stream
    check(L.lg_radix_sort_pairs_bounded(ka.data_ptr(), va.data_ptr(), kb.data_ptr(), vb.data_ptr(), n, n_dev.data_ptr(), 0, bits, temp.data_ptr(), tb, s), "sort")
    out_k, out_v = (kb, vb) if L.lg_radix_sort_num_passes(0, bits) % 2 else (ka, va)
    order = np.argsort(keys[:live], kind="stable")
    assert np.array_equal(host(out_k)[:live], keys[:live][order])
    assert np.array_equal(host(out_v)[:live], vals[:live][order])
    tr = torch.empty((1, ntiles + 2), dtype=torch.int32).cuda()
    check(L.lg_tile_range_bounded(out_k.data_ptr(), 1, n, n_dev.data_ptr(), ntiles, tr.data_ptr(), s), "range")
    from oracle import oracle as O
    assert np.array_equal(host(tr), O.tile_range(keys[:live][order][None], ntiles))


@pytest.mark.parametrize("V,W,H", [(1, 64, 64), (7, 1920, 1080), (300, 128, 64)])
def test_create_viewproj(F, oracle, V, W, H):
    """Learnable-camera matrices and their backward (GR/compact.cu:17-316), incl. the ordered fov-gradient sum for V > 256 lanes."""
    rng = np.random.default_rng(V)
    p7 = rng.standard_normal((V, 7)).astype(np.float32)
    p7[:, :4] *= rng.uniform(0.5, 2.0, (V, 1)).astype(np.float32)           # non-unit quaternions: same sphere projection as the reference
    fov = np.array([1.7], np.float32)
    outs = F.create_viewproj_forward(dev(p7), dev(fov), H, W, 0.01, 5000.0)
    refs = oracle.create_viewproj_forward(p7, fov, H, W, 0.01, 5000.0)
    for o, r in zip(outs, refs):
        assert_close(host(o), r, 1e-6, normalize=True)
    gv, gp, gvp = (rng.standard_normal((V, 4, 4)).astype(np.float32) for _ in range(3))
    g7, gf = F.create_viewproj_backward(dev(gv), dev(gp), dev(gvp), dev(p7), dev(fov), H, W, 0.01, 5000.0)
    r7, rf = oracle.create_viewproj_backward(gv, gp, gvp, p7, fov, H, W, 0.01, 5000.0)
    assert_close(host(g7), r7, 1e-5, normalize=True)
    assert_close(host(gf), rf, 1e-5, normalize=True)


def test_create_viewproj_autograd_wrapper(oracle):
    from litegs_amd.wrapper import CreateViewProj
    rng = np.random.default_rng(0)
    p7 = rng.standard_normal((3, 7)).astype(np.float32)
    fov = np.array([1.2], np.float32)
    tp, tf = dev(p7).requires_grad_(True), dev(fov).requires_grad_(True)
    view, proj, vp, planes = CreateViewProj.apply(tp, tf, 48, 96, 0.01, 100.0)
    assert not planes.requires_grad
    gvp = rng.standard_normal((3, 4, 4)).astype(np.float32)
    (vp * dev(gvp)).sum().backward()
    z = np.zeros((3, 4, 4), np.float32)
    r7, rf = oracle.create_viewproj_backward(z, z, gvp, p7, fov, 48, 96, 0.01, 100.0)
    assert_close(host(tp.grad), r7, 1e-5, normalize=True)
    assert_close(host(tf.grad), rf, 1e-5, normalize=True)


def test_cluster_aabb_matches_reference_fixture():
    """render.get_cluster_AABB (activated scale/rotation, as litegs/training/trainer.py:86 calls it) vs the reference's own
    litegs/scene/cluster.py output stored in the golden fixture."""
    import os
    from litegs_amd import render as R, synthetic as S
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_chain.npz")))
    xyz, scale, rot = (dev(S.cluster(gold[k], 128)) for k in ("cl_xyz", "cl_scale", "cl_rot"))
    origin, extend = R.get_cluster_AABB(xyz, scale, rot)
    assert_close(host(origin), gold["cl_origin"], 1e-5, normalize=True)
    assert_close(host(extend), gold["cl_extend"], 1e-5, normalize=True)
