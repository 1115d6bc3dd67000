"""End-to-end GPU parity: render_preprocess + render + backward (through autograd) + sparse Adam vs the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, case, oracle_forward, pinned_count

pytestmark = pytest.mark.gpu


def _gpu_params(c):
    return [torch.nn.Parameter(torch.from_numpy(p).cuda()) for p in c["params"]]


@pytest.mark.parametrize("name", ["small", "pad"])
def test_render_forward_backward_matches_oracle(oracle, name):
    from litegs_amd import render as R
    c = case(name)
    res = oracle_forward(name)
    H, W = c["H"], c["W"]
    pp = R.PipelineParams()
    params = _gpu_params(c)
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    vis_id, vis_num, xyz, scale, rot, color, opacity = R.render_preprocess(None, None, planes, view, *params, None, None, pp, c["degree"])
    assert int(vis_num.item()) == res.nvis
    assert np.array_equal(vis_id.cpu().numpy(), res.visible_chunkid)
    valid_length = vis_num * pp.cluster_size
    img, trans, depth, normal, prim_vis = R.render(view, proj, xyz, scale, rot, color, opacity, valid_length, None, None, c["degree"], (H, W), pp)
    assert int((prim_vis > 0).sum().item()) == int((res.alloc > 0).sum())

    rng = np.random.default_rng(4)
    w = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    (img * torch.from_numpy(w).cuda()).sum().backward()
    # oracle gradient: d_img on the padded image, masked where clamp(0,1) saturates from below (min(C,1) == 1 keeps gradient 1);
    # comparison under the two-part rule of tests/util.py
    from tests.util import compacted_grads, parity_image_and_gradients
    for p in params:
        assert p.grad.shape == p.shape, "CompactedTensor must claim the full parameter shape"
    like = oracle.render_backward(res, c["params"], c["view"], c["proj"], np.zeros_like(res.img), H, W, c["degree"])[0]
    parity_image_and_gradients(oracle, res, img.detach().cpu().numpy(), compacted_grads(params, res.nvis, like), c["params"], c["view"], c["proj"],
                               w, H, W, c["degree"])


def test_training_step_updates_only_visible_chunks(oracle):
    """fwd + bwd + SparseGaussianAdam.step on the GPU vs oracle gradients + oracle Adam."""
    from litegs_amd import render as R
    from litegs_amd import optimizer as Opt
    c = case("small")
    res = oracle_forward("small")
    H, W = c["H"], c["W"]
    pp = R.PipelineParams()
    params = _gpu_params(c)
    before = [p.detach().clone() for p in params]
    opt, sched = Opt.get_optimizer(*params, 1.0, Opt.OptimizationParams())
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    vis_id, vis_num, xyz, scale, rot, color, opacity = R.render_preprocess(None, None, planes, view, *params, None, None, pp, 3)
    img, *_ = R.render(view, proj, xyz, scale, rot, color, opacity, vis_num * 128, None, None, 3, (H, W), pp)
    rng = np.random.default_rng(4)
    w = rng.standard_normal(img.shape).astype(np.float32)
    (img * torch.from_numpy(w).cuda()).sum().backward()
    opt.step(vis_id, vis_num, None)
    opt.zero_grad(set_to_none=True)
    d_img = np.zeros_like(res.img)
    inside = (res.img[..., :H, :W] >= 0) & (res.img[..., :H, :W] <= 1)
    d_img[..., :H, :W] = w * inside
    (grads, _) = oracle.render_backward(res, c["params"], c["view"], c["proj"], d_img, H, W, 3)
    lrs = {g["name"]: g["lr"] for g in opt.param_groups}
    order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
    invisible = np.setdiff1d(np.arange(c["params"][0].shape[-2]), res.visible_chunkid)
    for p, b, g_ref, nm, raw in zip(params, before, grads, order, c["params"]):
        ref = raw.copy().reshape(-1, raw.shape[-2], raw.shape[-1])
        m = np.zeros_like(ref); v = np.zeros_like(ref)
        oracle.adam_chunk(ref, g_ref.reshape(-1, g_ref.shape[-2], g_ref.shape[-1]), m, v, res.visible_chunkid, res.nvis, lrs[nm])
        got = p.detach().cpu().numpy().reshape(ref.shape)
        # Adam's first step is lr * sign(g): compare the update direction where the oracle gradient is clearly non-zero
        big = np.abs(g_ref.reshape(-1, g_ref.shape[-2], g_ref.shape[-1])) > 1e-4 * np.abs(g_ref).max()
        upd_got = (got - raw.reshape(ref.shape))[:, res.visible_chunkid][big]
        upd_ref = (ref - raw.reshape(ref.shape))[:, res.visible_chunkid][big]
        wrong = np.abs(upd_got - upd_ref) > 1e-3 * lrs[nm]
        pinned_count(f"adam.{nm}", int(wrong.sum()), int(wrong.size), int(np.ceil(2e-3 * wrong.size)), atol=1e-3 * lrs[nm])
        assert wrong.mean() < 2e-3, nm
        assert np.array_equal(got[:, invisible], b.cpu().numpy().reshape(ref.shape)[:, invisible]), f"{nm}: invisible chunks must not move"
