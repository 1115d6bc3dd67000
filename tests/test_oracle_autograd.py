"""Pins the parts of the oracle the reference cannot execute on a CPU (blend forward/backward, chain backward ops,
activation) against an INDEPENDENT formulation: plain torch float64 + autograd on tiny cases."""
import numpy as np
import pytest
import torch

from litegs_amd import synthetic as S


def tiny_case():
    n, W, H, f = 700, 96, 64, 90.0
    params = S.make_scene(n, seed=3, scale_mult=1.2)
    view, proj, planes = S.make_camera(W, H, f, f, (1.8, -0.3, 0.8))
    return params, view, proj, planes, H, W


def torch_blend(packed, sorted_point, tile_start, H, W, TH, TW):
    """Dense per-tile front-to-back blend in float64 with autograd; decision masks are constants (as in the kernels)."""
    gx, gy = (W + TW - 1) // TW, (H + TH - 1) // TH
    Hp, Wp = gy * TH, gx * TW
    img = torch.zeros((3, Hp, Wp), dtype=torch.float64)
    trans = torch.ones((Hp, Wp), dtype=torch.float64)
    last = torch.zeros((Hp, Wp), dtype=torch.int64)
    for tile in range(1, gx * gy + 1):
        start, end = int(tile_start[tile]), int(tile_start[tile + 1])
        if start < 0 or start >= end:
            continue
        tx, ty = (tile - 1) % gx, (tile - 1) // gx
        ys, xs = torch.meshgrid(torch.arange(ty * TH, ty * TH + TH, dtype=torch.float64), torch.arange(tx * TW, tx * TW + TW, dtype=torch.float64), indexing="ij")
        T = torch.ones((TH, TW), dtype=torch.float64)
        C = torch.zeros((3, TH, TW), dtype=torch.float64)
        lc = torch.zeros((TH, TW), dtype=torch.int64)
        for i in range(start, end):
            r = packed[int(sorted_point[i])]
            active = (T > 1.0 / 8192).detach()
            if not active.any():
                break
            dx, dy = r[0] - xs, r[1] - ys
            power = -0.5 * (r[2] * dx * dx + (r[3] + r[10]) * dx * dy + r[4] * dy * dy)     # r[10] = ic10 (second off-diagonal)
            alpha = r[8] * torch.exp(power)
            valid = (active & (alpha >= 1.0 / 256)).detach()
            alpha = torch.where(alpha > 255.0 / 256, alpha * 0 + 255.0 / 256 + (alpha - alpha.detach()), alpha)   # clamp keeps the gradient (reference)
            alpha = alpha * valid
            lc = lc + active.long()
            w = T * alpha
            C = C + torch.stack([r[5], r[6], r[7]])[:, None, None] * w
            T = T * (1 - alpha)
        img[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW] = C
        trans[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW] = T
        last[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW] = lc
    return img, trans, last


def test_raster_forward_backward_vs_torch_autograd(oracle):
    params, view, proj, planes, H, W = tiny_case()
    res = oracle.render_forward(params, view, proj, planes, H, W, 3)
    assert res.n_instances > 2000 and res.last.max() > 20
    N = res.packed.shape[1]
    pk = torch.tensor(res.packed[0].astype(np.float64))
    cols = [pk[:, k].clone().requires_grad_(True) for k in range(9)]
    b10 = pk[:, 3].clone().requires_grad_(True)
    half_b = [c for c in cols]
    rec = [torch.stack([cols[0][i], cols[1][i], cols[2][i], cols[3][i], cols[4][i], cols[5][i], cols[6][i], cols[7][i], cols[8][i],
                        torch.zeros((), dtype=torch.float64), b10[i]]) for i in range(N)]
    img, trans, last = torch_blend(rec, res.sorted_point[0], res.tile_start[0], H, W, 8, 16)
    np.testing.assert_allclose(np.minimum(img.detach().numpy(), 1.0), res.img[0], atol=2e-5)
    np.testing.assert_allclose(trans.detach().numpy(), res.trans[0, 0], atol=2e-5)
    assert np.array_equal(last.numpy(), res.last[0, 0].astype(np.int64))

    rng = np.random.default_rng(0)
    d_img = rng.standard_normal(res.img.shape).astype(np.float32)
    (img * torch.tensor(d_img[0].astype(np.float64))).sum().backward()
    d_ndc, d_ic, d_color, d_opa, _ = oracle.raster_backward(res.sorted_point, res.tile_start, res.packed, res.trans, res.last, d_img, H, W, 8, 16)

    def rel(a, b):
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    # pixel -> ndc: px = (ndc+1)*0.5*W - 0.5
    assert rel(d_ndc[0, 0], cols[0].grad.numpy() * 0.5 * W) < 2e-4
    assert rel(d_ndc[0, 1], cols[1].grad.numpy() * 0.5 * H) < 2e-4
    assert rel(d_ic[0, 0, 0], cols[2].grad.numpy()) < 2e-4
    assert rel(d_ic[0, 0, 1], cols[3].grad.numpy()) < 2e-4          # each off-diagonal carries half of d/db
    assert rel(d_ic[0, 1, 0], b10.grad.numpy()) < 2e-4
    assert rel(d_ic[0, 1, 1], cols[4].grad.numpy()) < 2e-4
    for ch in range(3):
        assert rel(d_color[0, ch], cols[5 + ch].grad.numpy()) < 2e-4
    assert rel(d_opa[0], cols[8].grad.numpy()) < 2e-4


def test_chain_backward_ops_vs_autograd(oracle):
    rng = np.random.default_rng(1)
    N = 500
    f64 = lambda a: torch.tensor(a.astype(np.float64), requires_grad=True)
    # --- transform matrix
    quat = rng.standard_normal((4, N)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=0, keepdims=True)
    scale = (rng.random((3, N)) + 0.2).astype(np.float32)
    gT = rng.standard_normal((3, 3, N)).astype(np.float32)
    q, s = f64(quat), f64(scale)
    r, x, y, z = q[0], q[1], q[2], q[3]
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)]),
                     torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)]),
                     torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)])])
    T = R * s[:, None, :]
    np.testing.assert_allclose(oracle.transform_matrix_forward(quat, scale), T.detach().numpy(), atol=1e-6)
    (T * torch.tensor(gT.astype(np.float64))).sum().backward()
    gq, gs = oracle.transform_matrix_backward(gT, quat, scale)
    np.testing.assert_allclose(gq, q.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gs, s.grad.numpy(), rtol=1e-4, atol=1e-5)

    # --- mvp
    view, proj, _ = S.make_camera(640, 360, 500.0, 480.0, (1.0, -0.4, 3.0))
    world = np.concatenate([rng.standard_normal((3, N)), np.ones((1, N))]).astype(np.float32)
    w = f64(world)
    vp = (w.T @ torch.tensor(view[0].astype(np.float64))).T
    hom = (vp.T @ torch.tensor(proj[0].astype(np.float64))).T
    ndc = hom[:3] / hom[3:4]
    vp_o, ndc_o = oracle.mvp_forward(world, view, proj)
    np.testing.assert_allclose(vp_o[0], vp.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ndc_o[0, :3], ndc.detach().numpy(), rtol=1e-4, atol=1e-4)
    g_ndc = rng.standard_normal((1, 4, N)).astype(np.float32)
    g_view = rng.standard_normal((1, 4, N)).astype(np.float32)
    ((ndc * torch.tensor(g_ndc[0, :3].astype(np.float64))).sum() + (vp * torch.tensor(g_view[0].astype(np.float64))).sum()).backward()
    gw = oracle.mvp_backward(g_ndc, g_view, view, proj, vp_o)
    np.testing.assert_allclose(gw, w.grad.numpy(), rtol=2e-3, atol=2e-3)


def test_activation_vs_torch(oracle):
    rng = np.random.default_rng(2)
    params = S.make_scene(1000, seed=5)
    view, _, _ = S.make_camera(320, 200, 300.0, 300.0, (1.8, -0.3, 0.8))
    C, Sz = params[0].shape[-2:]
    ids = np.array([5, 2, 7], dtype=np.int64)
    pos, sc, rt, col, op = oracle.activate_forward(3, ids, 3, view, *params)
    xyz, scale, rot, sh0, shr, opa = [torch.tensor(p[..., ids, :].astype(np.float64), requires_grad=True) for p in params]
    np.testing.assert_allclose(sc, torch.exp(scale).detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(rt, torch.nn.functional.normalize(rot, dim=0).detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(op, torch.sigmoid(opa).detach().numpy(), atol=1e-6)
    cam = -view[0, 3, :3] @ view[0, :3, :3].T
    d = torch.nn.functional.normalize(xyz.detach() - torch.tensor(cam.astype(np.float64))[:, None, None], dim=0)
    # independent SH evaluation: the reference's own sh_to_rgb pins orc_sh2rgb in test_oracle_golden; reuse it here
    rgb = oracle.sh2rgb_forward(3, sh0.detach().numpy().reshape(1, 3, -1).astype(np.float32), shr.detach().numpy().reshape(15, 3, -1).astype(np.float32),
                                d.numpy().reshape(1, 3, -1).astype(np.float32))
    np.testing.assert_allclose(col.reshape(1, 3, -1), rgb, atol=2e-5)
    # backward: scale / rot chain rule vs autograd; opacity quirk g*sigmoid(x) (GR/compact.cu:952) checked explicitly
    g = [rng.standard_normal(a.shape).astype(np.float32) for a in (pos, sc, rt, col, op)]
    d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa = oracle.activate_backward(3, ids, 3, view, *params, *g)
    (torch.exp(scale) * torch.tensor(g[1].astype(np.float64))).sum().backward()
    (rot / torch.sqrt((rot * rot).sum(0, keepdim=True) + 1e-12) * torch.tensor(g[2].astype(np.float64))).sum().backward()
    np.testing.assert_allclose(d_scale, scale.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d_rot, rot.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(d_pos, g[0][:3], atol=0)
    np.testing.assert_allclose(d_opa, g[4] * torch.sigmoid(opa).detach().numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------- learnable cameras
def _torch_viewproj(p7, fov, H, W, zn, zf):
    q = p7[:, :4] / torch.sqrt((p7[:, :4] ** 2).sum(1, keepdim=True) + 1e-12)
    r, x, y, z = q.unbind(1)
    one, zero = torch.ones_like(r), torch.zeros_like(r)
    view = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y), zero], 1),
        torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x), zero], 1),
        torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y), zero], 1),
        torch.stack([p7[:, 4], p7[:, 5], p7[:, 6], one], 1)], 1)
    f = fov[0]
    proj = torch.zeros((4, 4), dtype=p7.dtype)
    proj = torch.stack([torch.stack([f, f * 0, f * 0, f * 0]), torch.stack([f * 0, f * W / H, f * 0, f * 0]),
                        torch.tensor([0, 0, zf / (zf - zn), 1.0], dtype=p7.dtype), torch.tensor([0, 0, -zf * zn / (zf - zn), 0], dtype=p7.dtype)])
    proj = proj[None].expand(p7.shape[0], 4, 4)
    return view, proj, view @ proj


@pytest.mark.parametrize("W,H", [(64, 64), (128, 64)])
def test_create_viewproj_backward_matches_autograd(oracle, W, H):
    """Integer aspect ratios only: for W/H non-integer the reference's fov gradient uses int division (compact.cu:268),
    covered by test_create_viewproj_backward_integer_aspect_quirk."""
    rng = np.random.default_rng(5)
    V = 5
    p7 = rng.standard_normal((V, 7)).astype(np.float32)
    p7[:, :4] /= np.linalg.norm(p7[:, :4], axis=1, keepdims=True)          # unit quaternions: reference drops the 1/|q| factor
    fov = np.array([1.3], np.float32)
    gv, gp, gvp = (rng.standard_normal((V, 4, 4)).astype(np.float32) for _ in range(3))
    view, proj, vp, planes = oracle.create_viewproj_forward(p7, fov, H, W, 0.01, 100.0)
    tp7, tfov = torch.tensor(p7, dtype=torch.float64, requires_grad=True), torch.tensor(fov, dtype=torch.float64, requires_grad=True)
    tview, tproj, tvp = _torch_viewproj(tp7, tfov, H, W, 0.01, 100.0)
    np.testing.assert_allclose(view, tview.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(proj, tproj.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vp, tvp.detach().numpy(), rtol=1e-5, atol=1e-5)
    loss = (tview * torch.tensor(gv)).sum() + (tproj * torch.tensor(gp)).sum() + (tvp * torch.tensor(gvp)).sum()
    loss.backward()
    g7, gf = oracle.create_viewproj_backward(gv, gp, gvp, p7, fov, H, W, 0.01, 100.0)
    np.testing.assert_allclose(g7, tp7.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gf, tfov.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_create_viewproj_backward_integer_aspect_quirk(oracle):
    """1920x1080: forward uses 16/9 for proj[1][1], backward multiplies its gradient by int(1920/1080) == 1."""
    p7 = np.array([[1, 0, 0, 0, 0, 0, 0]], np.float32)
    fov = np.array([1.0], np.float32)
    z = np.zeros((1, 4, 4), np.float32)
    gp = z.copy(); gp[0, 1, 1] = 1.0
    view, proj, vp, planes = oracle.create_viewproj_forward(p7, fov, 1080, 1920, 0.01, 100.0)
    assert abs(proj[0, 1, 1] - 1920 / 1080) < 1e-6
    _, gf = oracle.create_viewproj_backward(z, gp, z, p7, fov, 1080, 1920, 0.01, 100.0)
    assert gf[0] == 1.0


def test_tile_walk_drops_nan_cuts(oracle):
    """Two splats from the 3M @1080p case whose ellipse cut on the extreme tile line has discriminant ~ -1e-7 (sqrt -> NaN).
    The reference's device min()/max() drop the NaN operand (speedy_splat.cuh:105,115); a plain (a<b?a:b) would lose tiles."""
    ndc = np.array([[[-0.2990493178367615, 0.30606648325920105], [0.1544468104839325, 0.017400385811924934],
                     [0.9985971450805664, 0.9981240630149841], [1.0, 1.0]]], np.float32)
    inv = np.array([[[[0.05738087370991707, 0.013648195192217827], [0.06315372884273529, 0.016611842438578606]],
                     [[0.06315372884273529, 0.016611842438578606], [0.17696401476860046, 0.08646836876869202]]]], np.float32)
    op = np.array([[0.27023929357528687, 0.23289351165294647]], np.float32)
    vd = np.array([[7.118446350097656, 5.324782848358154]], np.float32)
    lu, rd, al = oracle.get_allocate_size(ndc, vd, inv, op, 1080, 1920, 8, 16)
    assert al.tolist() == [[5, 13]]


# ------------------------------------------------------------------------------------- exact tile binning
def _min_quadratic_on_rect(a, b, c, px, py, x0, x1, y0, y1):
    """min over the closed rectangle of a dx^2 + 2 b dx dy + c dy^2 (positive definite), float64, exact up to rounding:
    0 if the centre is inside, otherwise the minimum over the four edges (1-D quadratics clamped to the edge)."""
    if x0 <= px <= x1 and y0 <= py <= y1:
        return 0.0
    best = np.inf
    for y in (y0, y1):                      # horizontal edges: minimise over x
        dy = y - py
        x = np.clip(px - b * dy / a, x0, x1)
        dx = x - px
        best = min(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    for x in (x0, x1):                      # vertical edges: minimise over y
        dx = x - px
        y = np.clip(py - b * dx / c, y0, y1)
        dy = y - py
        best = min(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    return best


def test_accutile_emits_exactly_the_tiles_the_ellipse_touches(oracle):
    """The SpeedySplat/AccuTile walk restated in the oracle (GR/speedy_splat.cuh:16-149, GR/binning.cu:310-373) against an
    independent float64 geometric test: tile (tx,ty) is emitted iff its rectangle [tx*TW,(tx+1)*TW] x [ty*TH,(ty+1)*TH] (pixel-centre
    coordinates) intersects {x : (x-p)^T A (x-p) <= 2 ln(255 o)}.  Disagreements are only tolerated where the minimum of the
    quadratic over the rectangle is within 1e-3 (relative) of the threshold, or on the far open edge of a tile."""
    params, view, proj, planes, H, W = tiny_case()
    res = oracle.render_forward(params, view, proj, planes, H, W, 3)
    TH, TW = 8, 16
    gx, gy = (W + TW - 1) // TW, (H + TH - 1) // TH
    op = res.act[4][0]
    keys, vals = [a[0] for a in oracle.create_table(res.ndc, res.inv_cov, res.act[4], res.prefix, res.depth_sorted_index, H, W, TH, TW)[:2]]
    emitted = set(zip((keys - 1).tolist(), vals.tolist()))
    assert len(emitted) == res.n_instances > 1000
    checked = borderline = 0
    for i in np.nonzero(res.alloc[0] > 0)[0][:400]:
        a, b, c = (float(res.inv_cov[0, 0, 0, i]), float(res.inv_cov[0, 0, 1, i]), float(res.inv_cov[0, 1, 1, i]))
        px = (float(res.ndc[0, 0, i]) * 0.5 + 0.5) * W - 0.5
        py = (float(res.ndc[0, 1, i]) * 0.5 + 0.5) * H - 0.5
        t = 2.0 * np.log(float(op[i]) * 255.0)
        for ty in range(gy):
            for tx in range(gx):
                qmin = _min_quadratic_on_rect(a, b, c, px, py, tx * TW, (tx + 1) * TW, ty * TH, (ty + 1) * TH)
                want = qmin <= t
                got = (ty * gx + tx, int(i)) in emitted
                checked += 1
                if want != got:
                    # tolerated: threshold grazing, or the ellipse only touches the rectangle's far (open) boundary
                    qin = _min_quadratic_on_rect(a, b, c, px, py, tx * TW, (tx + 1) * TW - 1e-3, ty * TH, (ty + 1) * TH - 1e-3)
                    assert abs(qmin - t) <= 1e-3 * t or (qin > t) != want, (i, tx, ty, qmin, t, want, got)
                    borderline += 1
    assert checked > 5_000 and borderline <= 5e-3 * checked, (checked, borderline)
