"""The reference dispatches four tile shapes (GR/binning.cu:181-195, :422-436; GR/raster.cu:375-383): 8x16 (default, covered
everywhere else), 12x16, 16x16 and 8x8.  Each is checked here on both product paths against the CPU oracle: binning tables bit-exact,
image and all six parameter gradients to the float tolerance of the 8x16 tests."""
import numpy as np
import pytest
import torch

from tests.util import case, compacted_grads, parity_image_and_gradients

pytestmark = pytest.mark.gpu

NAMES = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]


@pytest.mark.parametrize("tile", [(12, 16), (16, 16), (8, 8)])
def test_tile_shapes_match_oracle(oracle, tile):
    from litegs_amd import fast, fused as F, render as R
    c = case("pad")                                          # image size not a multiple of any tile shape: padding rows/columns
    H, W = c["H"], c["W"]
    res = oracle.render_forward(c["params"], c["view"], c["proj"], c["planes"], H, W, c["degree"], tile=tile)
    assert res.n_instances > 1000
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # binning operators at this tile shape: counts, table, ranges -- integers, bit-exact
    op = res.act[4]
    vd = np.ascontiguousarray(res.view_pos[:, 2, :])
    _, _, al = F.get_allocate_size(dev(res.ndc), dev(vd), dev(res.inv_cov), dev(op), H, W, tile[0], tile[1], None)
    assert np.array_equal(al.cpu().numpy(), res.alloc)
    ks, vs = F.create_table(dev(res.ndc), dev(res.inv_cov), dev(op), dev(res.prefix), dev(res.depth_sorted_index), None, None, H, W, *tile)
    assert np.array_equal(ks.cpu().numpy(), res.sorted_tile) and np.array_equal(vs.cpu().numpy(), res.sorted_point)
    ntiles = ((H + tile[0] - 1) // tile[0]) * ((W + tile[1] - 1) // tile[1])
    assert np.array_equal(F.tileRange(ks, ntiles).cpu().numpy(), res.tile_start)

    rng = np.random.default_rng(7)
    w = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    like = oracle.render_backward(res, c["params"], c["view"], c["proj"], np.zeros_like(res.img), H, W, c["degree"], tile=tile)[0]
    view, proj, planes = [dev(x) for x in (c["view"], c["proj"], c["planes"])]
    imgs = []
    for mode in ("ops", "fused"):
        params = [torch.nn.Parameter(dev(p)) for p in c["params"]]
        if mode == "ops":
            pp = R.PipelineParams(tile_size=tile)
            vis_id, vis_num, xyz, scale, rot, color, opacity = R.render_preprocess(None, None, planes, view, *params, None, None, pp, c["degree"])
            img, *_ = R.render(view, proj, xyz, scale, rot, color, opacity, vis_num * pp.cluster_size, None, None, c["degree"], (H, W), pp)
        else:
            with torch.no_grad():
                origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
            rd = fast.FusedRenderer(1, H, W, tile=tile)
            img, vis_id, vis_num = rd.render(fast.CameraFrame(view, proj, planes, 0), origin, extend, *params, c["degree"])
        assert int(vis_num.item()) == res.nvis
        (img * dev(w)).sum().backward()
        parity_image_and_gradients(oracle, res, img.detach().cpu().numpy(), compacted_grads(params, res.nvis, like), c["params"], c["view"], c["proj"],
                                   w, H, W, c["degree"], tile=tile, tag=f"[{mode}]")
        imgs.append(img.detach().cpu().numpy())
    assert np.array_equal(imgs[0], imgs[1]), "executor and operator path must render the same image bit for bit"
