"""Depth-bound culling of the native executor (csrc/fused.hip): a revisited frame emits only the splats its tiles can reach, the image
stays bit-identical, and a violated bound is repaired by the gated fallback without any host decision."""
import numpy as np
import pytest
import torch

from litegs_amd import synthetic as S

from tests.util import noise_log

pytestmark = pytest.mark.gpu


def _scene(n=150_000, W=640, H=360, f=380.0):
    from litegs_amd import fast, render as R
    params = [torch.nn.Parameter(torch.from_numpy(p).cuda()) for p in S.make_scene(n, seed=5)]
    view, proj, planes = [torch.from_numpy(x).cuda() for x in S.make_camera(W, H, f, f, (1.6, -0.4, 1.1))]
    with torch.no_grad():
        origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
    cam = fast.CameraFrame(view, proj, planes, 0)
    return params, cam, origin, extend, H, W


def _render(rd, cam, origin, extend, params, w=None):
    for p in params:
        p.grad = None
    img, vis_id, vis_num = rd.render(cam, origin, extend, *params, 3)
    if w is not None:
        (img * w).sum().backward()
    torch.cuda.synchronize()
    return img.detach().clone()


def _bounds(rd, H, W, th=8, tw=16):
    """float view of the frame's current depth-bound block (layout: csrc/lg_tilewalk.h)"""
    gx, gy = -(-W // tw), -(-H // th)
    upper = sum((-(-gx // (1 << k))) * (-(-gy // (1 << k))) for k in range(1, 4))
    blk = rd.sched[0, rd.sched_cur[0]]
    return blk[:upper + gx * gy].view(torch.float32)


def _flags(rd):
    from litegs_amd._lib import lib
    ws1, N = rd.last_ws1
    off = lib().lg_fused_flags_offset(N)
    return ws1[off:off + 8].view(torch.int32).cpu().numpy()


def test_culled_visit_is_bit_identical_and_emits_less(depth_order_mode):
    from litegs_amd import fast
    params, cam, origin, extend, H, W = _scene()
    w = torch.from_numpy(np.random.default_rng(2).standard_normal((1, 3, H, W)).astype(np.float32)).cuda()
    rd = depth_order_mode(fast.FusedRenderer(1, H, W))
    rd.margin_fixed = 50                       # short lists at this scene size: the default 100 % margin reaches most of them
    rd.reset_feedback()
    img0 = _render(rd, cam, origin, extend, params, w)
    nvis = int(rd.fb_vis[0])
    g0 = [p.grad.compacted_values[..., :nvis, :].clone() for p in params]
    full = int(rd.fb_total[0])
    assert not rd.last_cull
    img1 = _render(rd, cam, origin, extend, params, w)
    assert rd.last_cull, "the second visit of the frame must run culled"
    culled = int(rd.fb_total[0])
    assert _flags(rd)[0] == 0, "no bound was violated: the fallback must not have run"
    assert torch.equal(img0, img1)
    assert culled < 0.95 * full, (culled, full)
    for a, p in zip(g0, params):
        b = p.grad.compacted_values[..., :nvis, :]
        assert torch.allclose(a, b, rtol=0, atol=2e-5 * float(a.abs().max()) + 1e-12)
    # ... and a third visit (bounds produced by a culled visit) again
    img2 = _render(rd, cam, origin, extend, params, w)
    assert rd.last_cull and _flags(rd)[0] == 0 and torch.equal(img0, img2)


@pytest.fixture(params=["global", "tile+scatter", "tile+radix"])
def depth_order_mode(request):
    """the three ways the executor builds the tile lists (csrc/fused.hip): depth sort of the splats + stable tile sort; per-tile depth sort
    behind the tile scatter (no sort over the instances); per-tile depth sort behind the stable tile radix sort"""
    mode, scatter = {"global": (0, True), "tile+scatter": (1, True), "tile+radix": (1, False)}[request.param]

    def apply(rd):
        rd.depth_order, rd.tile_scatter = mode, scatter
        return rd
    return apply


def test_violated_bounds_take_the_gated_fallback(depth_order_mode):
    from litegs_amd import fast
    params, cam, origin, extend, H, W = _scene()
    rd = depth_order_mode(fast.FusedRenderer(1, H, W))
    img0 = _render(rd, cam, origin, extend, params)
    full = int(rd.fb_total[0])
    _bounds(rd, H, W).mul_(0.2)                 # bounds far too tight: most tiles cannot saturate inside them
    img1 = _render(rd, cam, origin, extend, params)
    assert rd.last_cull
    assert _flags(rd)[0] == 1, "the culled run must have raised the fallback flag"
    assert _flags(rd)[1] == full and int(rd.fb_full[0]) == full, "the fallback rebuilds the full table"
    assert torch.equal(img0, img1)
    # the fallback left fresh bounds: the next visit is culled and clean again
    img2 = _render(rd, cam, origin, extend, params)
    assert rd.last_cull and _flags(rd)[0] == 0 and torch.equal(img0, img2)


def test_scene_change_between_visits_stays_exact(depth_order_mode):
    """opacities drop between two visits (tiles saturate deeper than predicted): whatever the culled visit decides, the image equals
    the one a fresh, unculled renderer produces"""
    from litegs_amd import fast
    params, cam, origin, extend, H, W = _scene()
    rd = depth_order_mode(fast.FusedRenderer(1, H, W))
    _render(rd, cam, origin, extend, params)
    for shift in (0.3, 1.0, 3.0):
        with torch.no_grad():
            params[5].sub_(shift)              # raw opacity (pre-sigmoid)
        img_c = _render(rd, cam, origin, extend, params)
        assert rd.last_cull
        ref = depth_order_mode(fast.FusedRenderer(1, H, W))
        ref.cull_enabled = False
        img_r = _render(ref, cam, origin, extend, params)
        assert torch.equal(img_c, img_r), f"shift {shift}: culled visit differs (fallback flag {_flags(rd)[0]})"


@pytest.mark.stochastic
def test_speculative_culling_replays_failed_steps():
    """Speculative mode (csrc/fused.hip "Speculative culling", litegs_amd/trainer.py): no gated repeat is enqueued; a violated bound poisons
    the fused backward + Adam launches from that step on, and the trainer replays those steps (the first unculled) when it notices.  With
    bounds sabotaged before two visits the replay must happen, every step's Adam must have run exactly once afterwards, and the result must
    agree with the non-speculative trainer (gated repeat) up to the run-to-run noise of the blend backward's float atomics."""
    from litegs_amd.trainer import SyntheticTrainer

    def run(speculative, drop_step=-1):
        tr = SyntheticTrainer(150_000, 640, 360, 380.0, n_frames=2, seed=5)
        tr.speculative = speculative
        rd = tr.renderer
        losses = []
        for i in range(14):
            if i in (6, 9):                                  # frame i % 2 was visited before: its next visit culls against these bounds
                torch.cuda.synchronize()
                k = i % 2
                gx, gy = -(-640 // 16), -(-360 // 8)
                upper = sum((-(-gx // (1 << q))) * (-(-gy // (1 << q))) for q in range(1, 4))
                rd.sched[k, rd.sched_cur[k]][:upper + gx * gy].view(torch.float32).mul_(0.2)
            if i == drop_step:                               # the defect this test exists to catch: one step's update is missing
                continue
            tr.step(i % 2)
            losses.append(tr.last["loss"])
        tr.flush()
        torch.cuda.synchronize()
        return tr, [float(l) for l in losses]

    ta, la = run(False)
    t_lost, _ = run(False, drop_step=12)                     # the same gated run with step 12 lost: the distance that must be told apart
    tb, lb = run(True)
    assert ta.renderer.fallbacks >= 1                        # the gated repeat really ran in the reference run
    assert tb.spec_replays >= 2, tb.spec_replays             # the failed step and at least the one enqueued behind it
    rb = tb.renderer
    assert not rb.poisoned() and int(rb.spec_poison.item()) == 0
    assert rb.applied_step() == 14                           # every step's Adam launch has run, the last one being step 14
    # Adam turns a last-bit gradient difference of a Gaussian with a near-zero gradient into a +-lr step, so the MAXIMUM difference between
    # two correct runs is already of the order of the parameter change; the MEAN absolute difference separates the cases: two correct runs
    # differ by the atomics-order noise (profiles/r04_spec_noise.log, 3 repetitions x 5 pairings: scale 4.6e-7 ... 4.5e-6, opacity 2.2e-6 ...
    # 2.3e-5 -- the noise itself moves 10x from run to run, so it is NOT used as the yardstick), a lost or doubled Adam step moves every
    # visible Gaussian by about one learning-rate step (scale 1.35e-3, opacity 6.7e-3: 300x the largest noise seen, and deterministic to
    # three digits).  The bound sits in the middle of that gap, in units of the lost-step distance measured in this very run.
    for pa, pl, pb in zip(ta.params, t_lost.params, tb.params):
        assert torch.isfinite(pb).all()
        d_lost = (pa.detach() - pl.detach()).abs().mean().item()
        d = (pa.detach() - pb.detach()).abs().mean().item()
        assert d_lost > 0
        noise_log(what="spec vs gated / lost step", ratio=d / d_lost, bound=0.06)
        assert d <= 0.06 * d_lost, (d, d_lost)              # sqrt(1/300) ~ 0.058: geometric middle of noise and lost step
    np.testing.assert_allclose(la, lb, rtol=2e-4)             # the loss of every step: what each step saw is what the gated run saw
    moved = max((p.detach() - torch.from_numpy(q).cuda()).abs().max().item() for p, q in zip(tb.params, S.make_scene(150_000, seed=5)))
    assert moved > 1e-2                                      # the 14 steps really changed the parameters
    np.testing.assert_allclose(la[:6], lb[:6], rtol=1e-4)    # before the first sabotage the two runs are the same computation
