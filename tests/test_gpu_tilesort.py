"""Per-tile depth sort (csrc/tilesort.hip): the table it leaves is the one a stable depth sort of the splats followed
by a stable tile sort produces -- checked (a) on synthetic tables with every list-length regime against numpy, (b) end to end: the
executor renders bit-identical images with the per-tile sort and with the global depth sort it replaces, on first visits, culled
visits and forced fallbacks."""
import numpy as np
import pytest
import torch

from tests.util import noise_log

pytestmark = pytest.mark.gpu


def _depth_key(d):
    u = d.view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000)


@pytest.mark.parametrize("case", ["mixed", "ties", "long"])
def test_tile_depth_sort_matches_stable_sorts(case):
    from litegs_amd import fused
    from litegs_amd._lib import check, lib
    rng = np.random.default_rng({"mixed": 1, "ties": 2, "long": 3}[case])
    if case == "long":
        lengths = [2049, 0, 4096, 5000, 1, 9001, 2048, 16385, 1]          # the table's LAST run has no closing entry (tileRange): keep it trivial
    else:
        lengths = [0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 190, 255, 256, 257, 511, 512, 513, 700, 1023, 1024, 1025, 2047, 2048, 0, 0, 5, 2100]
        lengths += list(rng.integers(0, 600, size=120)) + [1]
    ntiles = len(lengths) + 3                                        # the last tiles stay empty
    N = 40000
    depth = rng.uniform(0.01, 50.0, size=N).astype(np.float32)
    if case == "ties":
        depth = rng.choice(np.array([0.5, 1.0, 1.0000001, 7.25], dtype=np.float32), size=N)
    depth[:8] = np.array([-1.0, -0.0, 0.0, 1e-30, 3e38, -3e38, 2.5, 2.5], dtype=np.float32)
    keys, vals = [], []
    for t, n in enumerate(lengths):
        ids = np.sort(rng.choice(N, size=int(n), replace=False)) if n else np.zeros((0,), np.int64)
        if n >= 8 and t % 3 == 0:
            ids[:8] = np.arange(8)                                   # the special depths take part
            ids = np.unique(ids)
        keys.append(np.full((len(ids),), t + 1, np.int32)); vals.append(ids.astype(np.int32))
    keys, vals = np.concatenate(keys), np.concatenate(vals)
    L = len(keys)
    dev = torch.device("cuda", 0)
    tk, tv, pk = torch.from_numpy(keys[None]).to(dev), torch.from_numpy(vals[None].copy()).to(dev), torch.from_numpy(depth[None].copy()).to(dev)
    start = fused.tileRange(tk, ntiles)
    scratch = torch.full((1, L), 0x7fffffff, dtype=torch.int32, device=dev)
    check(lib().lg_tile_depth_sort(tv.data_ptr(), start.data_ptr(), pk.data_ptr(), 1, L, N, ntiles, scratch.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "tile_depth_sort")
    got = tv.cpu().numpy()[0]
    st = start.cpu().numpy()[0]
    off = 0
    for t, n in enumerate(lengths):
        n = int((keys == t + 1).sum())
        ids = vals[off:off + n]
        a, b = st[t + 1], st[t + 2]
        seen = (a >= 0 and b > a)
        if n == 0:
            off += n
            continue
        if not seen:                                                 # tileRange's convention: the last run has no closing entry -> not a list
            np.testing.assert_array_equal(got[off:off + n], ids)
            off += n
            continue
        assert (a, b) == (off, off + n)
        order = np.lexsort((ids, _depth_key(depth[ids])))            # primary: depth key, ties: id  == stable sort of ascending ids
        np.testing.assert_array_equal(got[off:off + n], ids[order], err_msg=f"tile {t + 1} n={n}")
        off += n


def _render_all(tr, frames, visits):
    out = []
    for v in range(visits):
        for k in frames:
            with torch.no_grad():
                img = tr.forward_only(k)
            out.append(img.clone())
    torch.cuda.synchronize()
    return out


def test_executor_is_bit_identical_with_and_without_the_splat_sort():
    """first visits (unculled, blocking sizes), culled revisits and the 16th-visit refresh: same images bit for bit in both modes"""
    from litegs_amd.trainer import SyntheticTrainer
    imgs = {}
    for mode in (1, 0):
        tr = SyntheticTrainer(60000, 640, 360, 500.0, n_frames=3, seed=5)
        tr.renderer.depth_order = mode
        imgs[mode] = _render_all(tr, range(3), 18)
        if mode == 1:
            assert tr.renderer.last_cull                          # the revisits really ran culled
    assert len(imgs[0]) == len(imgs[1]) == 54
    for a, b in zip(imgs[0], imgs[1]):
        assert torch.equal(a, b)


@pytest.mark.stochastic
def test_training_steps_agree_between_the_two_modes():
    """a few optimisation steps (forward, loss, blend backward, fused backward + Adam) with both orders: the blend backward's float
    atomics make parameters differ in the last bits run to run, so this is a tolerance check on top of the bit-exact forward test"""
    from litegs_amd.trainer import SyntheticTrainer
    res = {}
    for mode in (1, 0):
        tr = SyntheticTrainer(30000, 480, 270, 400.0, n_frames=2, seed=9)
        tr.renderer.depth_order = mode
        losses = [float(tr.step(i % 2).detach()) for i in range(12)]
        torch.cuda.synchronize()
        res[mode] = (losses, [p.detach().clone() for p in tr.params])
    # losses of the first steps agree closely; later steps (and the parameters) drift apart like any two runs of the SAME mode do, because
    # the blend backward's float atomics reorder the sums (tests/test_gpu_convergence.py measures that spread)
    a, b = np.asarray(res[0][0]), np.asarray(res[1][0])
    noise_log(what="loss, tile vs global order", rel_first4=float((np.abs(a - b) / np.abs(b))[:4].max()), rel_all=float((np.abs(a - b) / np.abs(b)).max()),
              bound_first4=2e-4, bound_all=2e-2)
    np.testing.assert_allclose(a[:4], b[:4], rtol=2e-4)
    np.testing.assert_allclose(a, b, rtol=2e-2)


@pytest.mark.parametrize("case", ["mixed", "ties"])
def test_tile_group_and_unordered_sort_equal_sorted_pipeline(case):
    """Grouping by tile WITHOUT a sort (lg_tile_group: per-key counts -> range table + cursors -> scatter) followed by the per-tile sort for
    lists in arbitrary order: same range table as tileRange of the sorted keys, same lists as lexsort by (depth key, id) -- with padding
    keys (0), empty tiles, every list-length regime of the per-tile sort and heavy ties in depth."""
    from litegs_amd import fused
    from litegs_amd._lib import check, lib
    rng = np.random.default_rng({"mixed": 11, "ties": 12}[case])
    lengths = [3, 0, 1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2048, 2049, 0, 0, 4500, 7] + list(rng.integers(0, 700, size=150)) + [0, 0, 9]
    ntiles = len(lengths) + 2                                        # trailing tiles stay empty
    N = 50000
    depth = rng.uniform(0.01, 50.0, size=N).astype(np.float32)
    if case == "ties":
        depth = rng.choice(np.array([0.5, 1.0, 1.0000001, 7.25, -2.0], dtype=np.float32), size=N)
    keys = np.concatenate([np.full((int(n),), t, np.int32) for t, n in enumerate(lengths)])          # key 0 (3 entries) = padding
    vals = np.concatenate([rng.choice(N, size=int(n), replace=False).astype(np.int32) for n in lengths])
    perm = rng.permutation(len(keys))
    keys_u, vals_u = keys[perm], vals[perm]
    L = len(keys)
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    tk, tv = torch.from_numpy(keys_u.copy()).to(dev), torch.from_numpy(vals_u.copy()).to(dev)
    start = torch.empty((1, ntiles + 2), dtype=torch.int32, device=dev)
    out = torch.full((1, L), -7, dtype=torch.int32, device=dev)
    temp = torch.empty((2 * (ntiles + 2),), dtype=torch.int32, device=dev)
    check(lib().lg_tile_group(tk.data_ptr(), tv.data_ptr(), L, ntiles, start.data_ptr(), out.data_ptr(), temp.data_ptr(), s), "tile_group")
    want_start = fused.tileRange(torch.from_numpy(np.sort(keys_u, kind="stable")[None]).to(dev), ntiles)
    assert torch.equal(start, want_start)
    got = out.cpu().numpy()[0]
    off = 0
    for t, n in enumerate(lengths):                                  # grouped: every segment holds exactly its key's values
        np.testing.assert_array_equal(np.sort(got[off:off + n]), np.sort(vals[keys == t]))
        off += int(n)
    pk = torch.from_numpy(depth[None].copy()).to(dev)
    scratch = torch.zeros((1, L), dtype=torch.int32, device=dev)
    check(lib().lg_tile_depth_sort_unordered(out.data_ptr(), start.data_ptr(), pk.data_ptr(), 1, L, N, ntiles, scratch.data_ptr(), s), "tile_depth_sort_unordered")
    got = out.cpu().numpy()[0]
    st = start.cpu().numpy()[0]
    off = 0
    for t, n in enumerate(lengths):
        n = int(n)
        ids = vals[keys == t]
        if t >= 1 and n >= 1 and st[t] >= 0 and st[t + 1] > st[t]:
            assert (st[t], st[t + 1]) == (off, off + n)
            order = np.lexsort((ids, _depth_key(depth[ids])))
            np.testing.assert_array_equal(got[off:off + n], ids[order], err_msg=f"tile {t} n={n}")
        off += n


def test_executor_tables_match_the_oracle_with_the_tile_scatter(oracle):
    """the executor's tile-instance table in per-tile-depth-sort mode with the tile scatter (no radix sort over the instances): tile ranges and
    the depth-ordered lists bit for bit against the oracle's reference pipeline (stable depth sort, emission, stable tile sort)"""
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
    import ctypes
    for scatter, validate in ((True, False), (False, False), (True, True), (False, True)):
        rd = fast.FusedRenderer(1, H, W)
        rd.depth_order, rd.tile_scatter, rd.validate_tables = 1, scatter, validate     # validate: the table checks find nothing to report
        cam = fast.CameraFrame(view, proj, planes, 0)
        with torch.no_grad():
            rd.render(cam, origin, extend, *params, c["degree"])
        torch.cuda.synchronize()
        rd.check_tables()
        ws2, table_len, N = rd.last_ws2
        o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), table_len, N, H, W, 8, 16)
        o_ts = L.lg_fused_tile_start_offset(table_len, N, H, W, 8, 16)
        ntiles = res.tile_start.shape[1] - 2
        ts = ws2[o_ts:o_ts + 4 * (ntiles + 2)].view(torch.int32).cpu().numpy()
        pts = ws2[o_pts:o_pts + 4 * res.n_instances].view(torch.int32).cpu().numpy()
        np.testing.assert_array_equal(ts, res.tile_start[0], err_msg=f"scatter={scatter}")
        np.testing.assert_array_equal(pts, res.sorted_point[0], err_msg=f"scatter={scatter}")
