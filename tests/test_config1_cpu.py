"""BASELINE.json configs[0]: 10k random Gaussians, one camera @400x400, rasterize forward + backward on the CPU (plumbing, no GPU).
The reference ships no CPU rasteriser, so the timed path is the oracle restatement; the test checks its
size-independent invariants and prints the timing the config asks for ("oracle fwd+bwd, timing reported")."""
import time

import numpy as np

from tests.util import case, oracle_forward


def test_config1_oracle_forward_backward(oracle):
    c = case("10k_400")
    res = oracle_forward("10k_400")                     # first call (cached for the other tests): includes building the C library
    H, W = c["H"], c["W"]
    t0 = time.perf_counter()
    res2 = oracle.render_forward(c["params"], c["view"], c["proj"], c["planes"], H, W, c["degree"])
    t_fwd = time.perf_counter() - t0
    assert np.array_equal(res2.img, res.img), "the oracle is deterministic"
    assert res.img.shape[-2:] == (400, 400) and np.isfinite(res.img).all()
    # binning invariants: sorted keys, every splat emitted allocate_size times, ranges partition [0, L)
    L = res.n_instances
    assert L > 0 and np.all(np.diff(res.sorted_tile[0].astype(np.int64)) >= 0)
    assert np.array_equal(np.bincount(res.sorted_point[0], minlength=res.alloc.shape[1]), res.alloc[0])
    ntiles = ((H + 7) // 8) * ((W + 15) // 16)
    starts = res.tile_start[0][1:ntiles + 1]
    assert res.tile_start[0][ntiles + 1] == L and np.all(np.diff(starts[starts >= 0]) > 0)
    # blend invariants: transmittance in (0, 1], colour bounded by the accumulated weight, `last` within the tile's list
    assert (res.trans > 0).all() and (res.trans <= 1).all()
    assert res.last.min() >= 0
    rng = np.random.default_rng(4)
    d_img = rng.standard_normal(res.img.shape).astype(np.float32)
    t0 = time.perf_counter()
    (grads, _) = oracle.render_backward(res, c["params"], c["view"], c["proj"], d_img, H, W, c["degree"])
    t_bwd = time.perf_counter() - t0
    for g, p in zip(grads, c["params"]):
        assert np.isfinite(g).all()
        assert g.shape[-1] == p.shape[-1]
    # linearity of the backward in d_img (a property of the algorithm, independent of size)
    (g2, _) = oracle.render_backward(res, c["params"], c["view"], c["proj"], 2.0 * d_img, H, W, c["degree"])
    for a, b in zip(grads, g2):
        np.testing.assert_allclose(2.0 * a, b, rtol=2e-5, atol=1e-6 * max(np.abs(b).max(), 1e-30))
    n = c["n"]
    print(f"\n[config1] 10k_400 oracle (CPU restatement; the reference ships no CPU path): forward {t_fwd * 1e3:.1f} ms "
          f"({n / t_fwd / 1e6:.3f} Msplats/s), backward {t_bwd * 1e3:.1f} ms, N_vis {res.nvis * 128}, instances {L}")
