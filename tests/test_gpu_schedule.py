"""Heaviest-first tile schedule of the blend kernels (raster.hip): the order is a permutation sorted by descending work, and the
schedule changes nothing but the order in which independent tiles run."""
import numpy as np
import pytest
import torch

from tests.util import case, oracle_forward

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_tile_order_is_a_descending_permutation():
    from litegs_amd._lib import lib, check
    L = lib()
    rng = np.random.default_rng(0)
    for ntiles in (1, 63, 1024, 16200, 40000):
        work = rng.integers(0, 3000, size=(2, ntiles + 1)).astype(np.int32)
        w = _dev(work)
        order = torch.empty((2, ntiles), dtype=torch.int32, device="cuda")
        check(L.lg_tile_order(w.data_ptr(), 2, ntiles, order.data_ptr(), torch.cuda.current_stream().cuda_stream), "tile_order")
        o = order.cpu().numpy()
        for v in range(2):
            assert np.array_equal(np.sort(o[v]), np.arange(1, ntiles + 1)), "not a permutation"
            key = np.minimum(work[v][o[v]], 1023)
            assert np.all(np.diff(key) <= 0), "not sorted by descending (clamped) work"


def test_tile_work_from_last_matches_numpy():
    from litegs_amd._lib import lib, check
    L = lib()
    rng = np.random.default_rng(1)
    for (H, W, th, tw) in ((200, 320, 8, 16), (141, 250, 16, 16), (64, 96, 8, 8), (100, 100, 12, 16)):
        gy, gx = -(-H // th), -(-W // tw)
        last = rng.integers(0, 500, size=(1, 1, gy * th, gx * tw)).astype(np.int16)
        work = torch.zeros((1, gx * gy + 1), dtype=torch.int32, device="cuda")
        check(L.lg_tile_work_from_last(_dev(last).data_ptr(), 1, H, W, th, tw, work.data_ptr(), torch.cuda.current_stream().cuda_stream), "tile_work")
        ref = last[0, 0].reshape(gy, th, gx, tw).max(axis=(1, 3)).reshape(-1)
        assert np.array_equal(work.cpu().numpy()[0, 1:], ref)


def test_schedule_does_not_change_results(oracle):
    """blend forward + backward with and without the schedule: image bit-identical, gradients equal up to the order of the atomics"""
    from litegs_amd._lib import lib, check
    from litegs_amd import fused as F
    res = oracle_forward("small")
    c = case("small")
    H, W = c["H"], c["W"]
    L = lib()
    sp, ts = _dev(res.sorted_point), _dev(res.tile_start)
    pos, sc, rt, col, op = res.act
    pk = F.rasterize_forward(sp, ts, _dev(res.ndc), _dev(res.inv_cov), _dev(col), _dev(op), None, H, W, 8, 16, False, False, False)[4]
    outs = []
    for use in (1, 0):
        check(L.lg_set_tuning(4, use), "tuning")
        img, trans, depth, last, fc, fw = F.rasterize_forward_packed(sp, ts, pk, None, H, W, 8, 16, False, False, False)
        d_img = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(img.shape)).astype(np.float32)).cuda()
        g = F.rasterize_backward(sp, ts, pk, None, trans, last, d_img, None, None, None, H, W, 8, 16, False)
        outs.append((img.cpu().numpy(), [x.cpu().numpy() for x in g[:4]]))
    check(L.lg_set_tuning(4, 1), "tuning")
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5 * max(np.abs(b).max(), 1e-30))
