"""Seeded synthetic Gaussian clouds and pinhole cameras.

Conventions follow the reference (litegs/data.py:13): row-major, row-vector matrices,
``view_pos = [x y z 1] @ V``; ``V = [[R, t],[0, 1]]^T``; ``proj`` as litegs/data.py:42-46
(near 0.01, far 5000); frustum planes as litegs/data.py:139-176; Gaussians are Morton-sorted
(litegs/scene/point.py:29-76) and chunked by 128 (litegs/scene/cluster.py:7-21) so chunk-level
frustum culling is meaningful.  Everything is numpy float32 on the host; callers move it to HBM.
"""
from __future__ import annotations

import math

import numpy as np


def morton_order(xyz: np.ndarray, bits: int = 21) -> np.ndarray:
    """Stable argsort of 3x21-bit Morton codes of xyz[3,N] (litegs/scene/point.py:29-76)."""
    mn = xyz.min(axis=1, keepdims=True)
    mx = xyz.max(axis=1, keepdims=True)
    scale = (1 << bits) - 1
    q = ((xyz - mn) / np.maximum(mx - mn, 1e-12) * scale).astype(np.int64).clip(0, scale)
    codes = np.zeros(xyz.shape[1], dtype=np.int64)
    for i in range(bits):
        codes |= (((q[0] >> i) & 1) << (3 * i)) | (((q[1] >> i) & 1) << (3 * i + 1)) | (((q[2] >> i) & 1) << (3 * i + 2))
    return np.argsort(codes, kind="stable")


def cluster(t: np.ndarray, chunk: int = 128) -> np.ndarray:
    """[..., N] -> [..., chunks, chunk]; the tail is padded by repeating the last entries
    (litegs/scene/cluster.py:14-18)."""
    n = t.shape[-1]
    if n % chunk:
        pad = chunk - n % chunk
        src = n - 1 - (np.arange(pad)[::-1] % n)          # the last `pad` entries; cyclic when the cloud is smaller than the padding
        t = np.concatenate([t, t[..., src]], axis=-1)
    return np.ascontiguousarray(t.reshape(*t.shape[:-1], t.shape[-1] // chunk, chunk))


def make_scene(n: int, seed: int = 0, sh_degree: int = 3, radius: float = 4.0, chunk: int = 128,
               scale_mult: float = 0.6):
    """Raw (pre-activation) parameters, chunked:
    xyz[3,C,S] scale[3,C,S] rot[4,C,S] sh_0[1,3,C,S] sh_rest[(d+1)^2-1,3,C,S] opacity[1,C,S]."""
    rng = np.random.default_rng(seed)
    xyz = ((rng.random((3, n), dtype=np.float32) * 2 - 1) * radius).astype(np.float32)
    mu = math.log(scale_mult * radius * n ** (-1.0 / 3.0))
    scale = (mu + 0.5 * rng.standard_normal((3, n), dtype=np.float32)).astype(np.float32)   # log of the axis length
    rot = rng.standard_normal((4, n), dtype=np.float32)
    o = rng.random((1, n), dtype=np.float32) * 0.9 + 0.05
    opacity = np.log(o / (1 - o)).astype(np.float32)
    sh_0 = (rng.standard_normal((1, 3, n), dtype=np.float32) * 0.5).astype(np.float32)
    nrest = (sh_degree + 1) ** 2 - 1
    sh_rest = (rng.standard_normal((max(nrest, 1), 3, n), dtype=np.float32) * 0.05).astype(np.float32)
    if nrest == 0:
        sh_rest = sh_rest[:0]
    order = morton_order(xyz)
    out = [cluster(np.ascontiguousarray(a[..., order]), chunk) for a in (xyz, scale, rot, sh_0, sh_rest, opacity)]
    return tuple(out)


def perturb(scene, seed: int, amount: float = 1.0):
    """student = teacher + seeded noise in raw parameter space (positions 2 % of the radius, 25 % axis lengths, colours, opacity; the SH
    rest starts at zero) -- the start of the teacher -> student training runs (tests/convergence*.py, bench.py's training_state)."""
    rng = np.random.default_rng(seed)
    xyz, scale, rot, sh0, shr, opa = [np.array(a, copy=True) for a in scene]
    xyz += amount * 0.08 * rng.standard_normal(xyz.shape).astype(np.float32)
    scale += amount * 0.25 * rng.standard_normal(scale.shape).astype(np.float32)
    rot += amount * 0.2 * rng.standard_normal(rot.shape).astype(np.float32)
    sh0 += amount * 0.4 * rng.standard_normal(sh0.shape).astype(np.float32)
    shr = shr * 0.0
    opa += amount * 0.7 * rng.standard_normal(opa.shape).astype(np.float32)
    return xyz, scale, rot, sh0, shr.astype(np.float32), opa


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def look_at(cam_pos, target, up=(0.0, -1.0, 0.0)):
    """World->camera rotation R and translation t (x_c = R x_w + t), camera looks down +z, y down (COLMAP)."""
    cam_pos = np.asarray(cam_pos, np.float64)
    fwd = np.asarray(target, np.float64) - cam_pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)
    t = -R @ cam_pos
    return R, t


def frustum_planes(view: np.ndarray, proj: np.ndarray) -> np.ndarray:
    """Six frustum planes [6,4] (left, right, bottom, top, near, far) from the columns of view@proj (litegs/data.py:139-176)."""
    vp = view @ proj
    pl = np.zeros((6, 4), np.float32)
    pl[0] = vp[:, 3] + vp[:, 0]
    pl[1] = vp[:, 3] - vp[:, 0]
    pl[2] = vp[:, 3] + vp[:, 1]
    pl[3] = vp[:, 3] - vp[:, 1]
    pl[4] = vp[:, 2]
    pl[5] = vp[:, 3] - vp[:, 2]
    return pl


def pinhole_proj(width: int, height: int, fx: float, fy: float, z_near: float = 0.01, z_far: float = 5000.0) -> np.ndarray:
    """Row-vector projection matrix of litegs/data.py:36-46."""
    a = fx / (width * 0.5)
    b = fy / (height * 0.5)
    return np.array([[a, 0, 0, 0], [0, b, 0, 0],
                     [0, 0, z_far / (z_far - z_near), -z_far * z_near / (z_far - z_near)],
                     [0, 0, 1, 0]], dtype=np.float32).T


def make_camera(width: int, height: int, fx: float, fy: float, cam_pos, target=(0, 0, 0),
                z_near: float = 0.01, z_far: float = 5000.0):
    """-> view_matrix[1,4,4], proj_matrix[1,4,4], frustumplane[1,6,4] (float32)."""
    R, t = look_at(cam_pos, target)
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    view = Rt.T.astype(np.float32)                       # litegs/utils/__init__.py:38-43 + data.py:77
    proj = pinhole_proj(width, height, fx, fy, z_near, z_far)
    return view[None].copy(), proj[None].copy(), frustum_planes(view, proj)[None].copy()


def orbit_cameras(count: int, width: int, height: int, fx: float, fy: float, radius: float,
                  elevation_deg: float = 15.0, target=(0, 0, 0), phase: float = 0.0):
    """`count` equally spaced azimuths on a circle of `radius` at `elevation_deg`."""
    cams = []
    el = math.radians(elevation_deg)
    for k in range(count):
        az = phase + 2 * math.pi * k / count
        pos = (radius * math.cos(el) * math.cos(az), -radius * math.sin(el), radius * math.cos(el) * math.sin(az))
        cams.append(make_camera(width, height, fx, fy, pos, target))
    return cams


# the BASELINE.json configurations (name -> (n_gaussians, width, height, focal))
CONFIGS = {
    "10k_400": (10_000, 400, 400, 400.0),
    "500k_1080p": (500_000, 1920, 1080, 1200.0),
    "3m_1080p": (3_000_000, 1920, 1080, 1200.0),
    "10m_1600x1200": (10_000_000, 1600, 1200, 1100.0),
}
# not in BASELINE.json: an intermediate size used to place the switch between the two depth-order modes (tools/margin_ab.py)
CONFIGS["6m_1080p"] = (6_000_000, 1920, 1080, 1200.0)
