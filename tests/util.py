"""Shared helpers for the parity tests: seeded cases, the oracle pipeline (cached) and comparison rules."""
from __future__ import annotations

import functools

import numpy as np

from litegs_amd import synthetic as S

# Tolerances.  north_star: images and gradients within 1e-4 (fp32).  Gradients span many decades, so they are
# compared after normalising each tensor by its max-abs.  The blend has measure-zero decision
# thresholds (alpha >= 1/256, T > 1/8192, tile membership): a 1-ulp difference in exp() can flip one decision
# for one (pixel, splat) pair, which moves that pixel by up to c/256.  Such flips are rare (<= FLIP_FRAC of the
# elements) and are bounded, not ignored: flipped elements must still be within FLIP_ATOL.
ATOL = 1e-4
FLIP_FRAC = 2e-5
FLIP_ATOL = 2e-2


def assert_close(got, ref, atol=ATOL, flip_frac=0.0, flip_atol=FLIP_ATOL, normalize=False, name=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    scale = 1.0
    if normalize:
        scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref) / scale
    bad = err > atol
    nbad = int(bad.sum())
    allowed = int(np.ceil(flip_frac * err.size))
    assert nbad <= allowed, f"{name}: {nbad} elements exceed {atol} (allowed {allowed}); max err {err.max():.3e}"
    if nbad:
        assert err.max() <= flip_atol, f"{name}: flipped element error {err.max():.3e} > {flip_atol}"
    return float(err.max())


@functools.lru_cache(maxsize=8)
def case(name: str = "small", seed: int = 0):
    """-> dict(params, view, proj, planes, H, W, degree)."""
    if name == "small":
        n, W, H, f = 6000, 320, 200, 300.0
    elif name == "10k_400":
        n, W, H, f = S.CONFIGS["10k_400"]
    elif name == "pad":          # image size not a multiple of the tile: exercises padded tiles
        n, W, H, f = 3000, 250, 141, 260.0
    elif name in S.CONFIGS:      # BASELINE.json sizes (500k_1080p, 3m_1080p): full-size parity, GPU tests only
        n, W, H, f = S.CONFIGS[name]
    else:
        raise KeyError(name)
    params = S.make_scene(n, seed=seed)
    view, proj, planes = S.make_camera(W, H, f, f, (0.5 * 4 * 0.9, -0.3, 0.5 * 4 * 0.4))
    return dict(params=params, view=view, proj=proj, planes=planes, H=H, W=W, degree=3, n=n)


@functools.lru_cache(maxsize=8)
def oracle_forward(name: str = "small", seed: int = 0, stat: bool = False):
    from oracle import oracle as O
    c = case(name, seed)
    return O.render_forward(c["params"], c["view"], c["proj"], c["planes"], c["H"], c["W"], c["degree"], enable_stat=stat)


def d_img_for(res, seed=1):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(res.img.shape).astype(np.float32)
