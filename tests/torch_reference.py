"""An INDEPENDENT differentiable formulation of the hot path in plain torch + autograd (test infrastructure, not product).

Dense compositing: every Gaussian against every pixel, no tiles lists, no sort of (tile,depth) keys -- the tile structure of the
reference only enters as a membership mask (tile rectangle intersects the 1/255 ellipse, the geometric statement of AccuTile,
GR/speedy_splat.cuh:16-149).  Everything continuous is left to autograd, with the reference's four deliberate deviations from the
plain chain rule stated explicitly:
  * no gradient through the ray-space Jacobian or the SH view direction (wrapper.py:481-524 passes none),
  * d sigmoid = g * sigmoid(x) (GR/compact.cu:952),
  * the alpha clamp min(255/256, .) passes its gradient (GR/raster.cu:700-720),
  * non-finite gradients of the 2x2 covariance are zeroed (wrapper.py:591).
Decision masks (visible, alpha >= 1/256, T > 1/8192) are constants of the backward, as in the kernels.

Pinned against the oracle (forward image and the six parameter gradients) by tests/test_torch_reference.py; used by
tests/convergence.py as the third training path.
"""
from __future__ import annotations

import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199


class _OpacityAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.sigmoid(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g * y


def sh_basis(d: torch.Tensor, degree: int) -> torch.Tensor:
    """d [3,N] unit directions -> [(degree+1)^2, N]; constants and signs of GR/compact.cu:553-653."""
    x, y, z = d[0], d[1], d[2]
    b = [torch.full_like(x, C0)]
    if degree > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.31539156525252005 * (2 * zz - xx - yy),
              -1.0925484305920792 * xz, 0.5462742152960396 * (xx - yy)]
        if degree > 2:
            b += [-0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z, -0.4570457994644658 * y * (4 * zz - xx - yy),
                  0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy), -0.4570457994644658 * x * (4 * zz - xx - yy),
                  1.445305721320277 * z * (xx - yy), -0.5900435899266435 * x * (xx - 3 * yy)]
    return torch.stack(b)


def _min_quadratic_on_rects(a, b, c, px, py, x0, x1, y0, y1):
    """min over closed rectangles of a dx^2 + 2 b dx dy + c dy^2; a,b,c,px,py [N,1], rect bounds [1,T] -> [N,T]."""
    inside = (px >= x0) & (px <= x1) & (py >= y0) & (py <= y1)
    best = torch.full((a.shape[0], x0.shape[1]), float("inf"), dtype=a.dtype, device=a.device)
    for y in (y0, y1):
        dy = y - py
        x = torch.minimum(torch.maximum(px - b * dy / a, x0), x1)
        dx = x - px
        best = torch.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    for x in (x0, x1):
        dx = x - px
        y = torch.minimum(torch.maximum(py - b * dx / c, y0), y1)
        dy = y - py
        best = torch.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    return torch.where(inside, torch.zeros_like(best), best)


def project(params, view, proj, H, W, degree=3):
    """Activation + projection chain for one camera.  params: the six raw chunked tensors; view/proj [4,4] row-vector matrices.
    Returns pixel position [2,N], conic (a,b,c) [3,N], colour [3,N], opacity [N], view depth [N], visible [N] (bool)."""
    xyz, scale, rot, sh0, shr, opa = params
    N = xyz.shape[-2] * xyz.shape[-1]
    dt = xyz.dtype
    V, P = view.to(dt), proj.to(dt)
    pos = xyz.reshape(3, N)
    sc = torch.exp(scale.reshape(3, N))
    q = rot.reshape(4, N)
    q = q / torch.sqrt((q * q).sum(0, keepdim=True) + 1e-12)
    op = _OpacityAct.apply(opa.reshape(N))

    # colour: SH in the direction camera centre -> Gaussian, direction is a constant of the backward
    cam = -(V[3, :3] @ V[:3, :3].T)
    d = pos.detach() - cam[:, None]
    d = d / torch.sqrt((d * d).sum(0, keepdim=True) + 1e-12)
    basis = sh_basis(d, degree)                                                   # [K,N]
    sh = torch.cat([sh0.reshape(1, 3, N), shr.reshape(-1, 3, N)[: basis.shape[0] - 1]], 0)
    color = (basis[:, None, :] * sh).sum(0) + 0.5                                 # [3,N]

    # mvp (GR/transform.cu:398-436)
    w = torch.cat([pos, torch.ones((1, N), dtype=dt, device=pos.device)], 0)
    v = V.T @ w                                                                    # row-vector convention: v = w^T V
    h = P.T @ v
    iw = torch.where(h[3].abs() > 1e-12, 1.0 / h[3], torch.zeros_like(h[3]))
    ndc = h[:3] * iw
    pix = torch.stack([(ndc[0] + 1) * 0.5 * W - 0.5, (ndc[1] + 1) * 0.5 * H - 0.5])

    # T = R . diag(s) rows (GR/transform.cu:106-125)
    r, x, y, z = q[0], q[1], q[2], q[3]
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)]),
                     torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)]),
                     torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)])])      # [3,3,N]
    T = R * sc[:, None, :]

    # ray-space Jacobian (GR/transform.cu:36-50), constant of the backward
    with torch.no_grad():
        tx, ty, tz = v[0], v[1], v[2]
        fx, fy = P[0, 0] * W * 0.5, P[1, 1] * H * 0.5
        lx, ly = tz / P[0, 0] * 1.3, tz / P[1, 1] * 1.3
        tx = torch.maximum(torch.minimum(tx, lx), -lx)
        ty = torch.maximum(torch.minimum(ty, ly), -ly)
        rz = 1.0 / torch.clamp(tz, min=1e-2)
        zero = torch.zeros_like(rz)
        J = torch.stack([torch.stack([fx * rz, zero]), torch.stack([zero, fy * rz]),
                         torch.stack([-fx * tx * rz * rz, -fy * ty * rz * rz])])                               # [3,2,N]
        VJ = torch.einsum("rk,kcn->rcn", V[:3, :3], J)
    M = torch.einsum("rkn,kcn->rcn", T, VJ)                                                                    # [3,2,N]
    cov = torch.einsum("kan,kbn->abn", M, M) + 0.3 * torch.eye(2, dtype=dt, device=pos.device)[:, :, None]
    if cov.requires_grad:
        cov.register_hook(lambda g: torch.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0))
    m00, m01, m10, m11 = cov[0, 0], cov[0, 1], cov[1, 0], cov[1, 1]
    det = m00 * m11 - m01 * m10
    det1 = (m00 - m01) * (m11 - m01) + m01 * (m00 + m11 - 2 * m01)
    det = torch.where(det.abs() < (1e-5 * m01 * m10).abs(), det1, det)
    det = torch.where(det.abs() < 1e-9, torch.full_like(det, 1e-9), det)
    a, b01, b10, c = m11 / det, -m01 / det, -m10 / det, m00 / det

    with torch.no_grad():
        vis = ~((ndc[0] < -1.3) | (ndc[0] > 1.3) | (ndc[1] < -1.3) | (ndc[1] > 1.3) | (v[2] <= 0.2) | (op < 1.0 / 255))
        vis &= (a > 0) & (c > 0) & (b01 * b01 - a * c < 0)
    return pix, (a, b01, b10, c), color, op, v[2], vis


def render(params, view, proj, H, W, degree=3, tile=(8, 16), chunk_ids=None):
    """[3,H,W] image of one camera (min(C,1) with a straight-through gradient, as the blend backward does not mask it).
    chunk_ids: the chunks that survived frustum culling (None = all)."""
    TH, TW = tile
    pix, (a, b01, b10, c), color, op, depth, vis = project(params, view, proj, H, W, degree)
    dev, dt = pix.device, pix.dtype
    gx, gy = (W + TW - 1) // TW, (H + TH - 1) // TH
    Hp, Wp = gy * TH, gx * TW
    if chunk_ids is not None:
        keep = torch.zeros(params[0].shape[-2], dtype=torch.bool, device=dev)
        keep[torch.as_tensor(chunk_ids, device=dev).long()] = True
        vis = vis & keep[:, None].expand(-1, params[0].shape[-1]).reshape(-1)
    order = torch.sort(depth.detach(), stable=True).indices
    order = order[vis[order]]
    px, py = pix[0][order], pix[1][order]
    a, b01, b10, c, op, color = a[order], b01[order], b10[order], c[order], op[order], color[:, order]

    # tile membership: rectangle [tx*TW,(tx+1)*TW] x [ty*TH,(ty+1)*TH] (pixel-centre coordinates) meets the 1/255 ellipse
    with torch.no_grad():
        tix = torch.arange(gx, device=dev, dtype=dt).repeat(gy)
        tiy = torch.arange(gy, device=dev, dtype=dt).repeat_interleave(gx)
        x0, x1, y0, y1 = (tix * TW)[None], (tix * TW + TW)[None], (tiy * TH)[None], (tiy * TH + TH)[None]
        thr = 2.0 * torch.log(op * 255.0)
        qmin = _min_quadratic_on_rects(a[:, None], (0.5 * (b01 + b10))[:, None], c[:, None], px[:, None], py[:, None], x0, x1, y0, y1)
        in_tile = qmin <= thr[:, None]                                              # [n, tiles]
        ys, xs = torch.meshgrid(torch.arange(Hp, device=dev), torch.arange(Wp, device=dev), indexing="ij")
        tile_of_pixel = ((ys // TH) * gx + xs // TW).reshape(-1)
        X, Y = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)

    dx, dy = px[:, None] - X[None], py[:, None] - Y[None]                           # [n, P]
    power = -0.5 * (a[:, None] * dx * dx + c[:, None] * dy * dy) - 0.5 * (b01 + b10)[:, None] * dx * dy
    alpha = op[:, None] * torch.exp(power)
    with torch.no_grad():
        ok = in_tile[:, tile_of_pixel] & (alpha >= 1.0 / 256)
    alpha = alpha + (torch.clamp(alpha, max=255.0 / 256) - alpha).detach()
    alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha
    T_incl = torch.cumprod(one_minus, 0)
    T_before = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], 0)
    with torch.no_grad():
        active = T_before > 1.0 / 8192
    wgt = torch.where(active, alpha * T_before, torch.zeros_like(alpha))
    C = (color[:, :, None] * wgt[None]).sum(1).reshape(3, Hp, Wp)
    C = C + (torch.clamp(C, max=1.0) - C).detach()
    return C[:, :H, :W]


class PlainAdam:
    """Adam without bias correction over all rows (GR/compact.cu:333-342); with every chunk visible this is the sparse update."""

    def __init__(self, params, lrs, b1=0.9, b2=0.999, eps=1e-15):
        self.params, self.lrs, self.b1, self.b2, self.eps = params, list(lrs), b1, b2, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        for p, m, v, lr in zip(self.params, self.m, self.v, self.lrs):
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.add_(-lr * m / (v.sqrt() + self.eps))
            p.grad = None


def psnr(img: torch.Tensor, gt: torch.Tensor) -> float:
    mse = ((img.clamp(0, 1) - gt) ** 2).mean().item()
    return 10.0 * math.log10(1.0 / max(mse, 1e-12))
