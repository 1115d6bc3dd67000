"""The per-tile depth sort's network and its three list-length regimes, emulated on the host: litegs_amd/csrc/lg_tilesort_body.h is
written so that the same text compiles as device code (tilesort.hip) and as a sequential C++ program (tests/host/bitonic_check.cpp),
which compares it with std::stable_sort for ≈2100 lists (every length 2..700, chunk edges up to 20 000, duplicate-heavy depths, the
wave radix regime at every length, arbitrary arrival orders)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tile_sort_body_matches_stable_sort_on_the_host(tmp_path):
    exe = str(tmp_path / "bitonic_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "litegs_amd", "csrc"), os.path.join(ROOT, "tests", "host", "bitonic_check.cpp"),
                    "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_per_tile_sort_of_id_ordered_emission_equals_the_reference_pipeline():
    """The equivalence the per-tile depth sort rests on, on the oracle's tables (no GPU): stable depth sort of the splats + emission in
    that order + stable tile sort (the reference's pipeline)  ==  emission in ascending id order + stable tile sort + per-tile sort by
    (depth key, id).  Duplicated depths are forced so that the tie-break matters."""
    import numpy as np
    from oracle import oracle as O
    from tests.util import case
    c = case("small")
    res = O.render_forward(c["params"], c["view"], c["proj"], c["planes"], c["H"], c["W"], c["degree"])
    H, W = c["H"], c["W"]
    ndc, inv, op = res.ndc, res.inv_cov, res.act[4]
    depth = np.ascontiguousarray(res.view_pos[:, 2, :]).copy()
    N = depth.shape[1]
    rng = np.random.default_rng(0)
    dup = rng.choice(N, size=N // 3, replace=False)                 # a third of the splats share their depth with another one
    depth[0, dup] = depth[0, rng.choice(N, size=N // 3)]
    _, _, alloc = O.get_allocate_size(ndc, depth, inv, op, H, W, 8, 16)
    # reference pipeline
    dsi = np.argsort(depth, axis=-1, kind="stable").astype(np.int64)
    prefix = np.cumsum(np.take_along_axis(alloc, dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
    ks_ref, vs_ref, *_ = O.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
    # id-ordered emission, stable tile sort (inside create_table), then the per-tile sort
    ident = np.arange(N, dtype=np.int64)[None]
    prefix_id = np.cumsum(alloc, axis=-1, dtype=np.int64).astype(np.int32)
    ks, vs, *_ = O.create_table(ndc, inv, op, prefix_id, ident, H, W, 8, 16)
    np.testing.assert_array_equal(ks, ks_ref)                        # same multiset of tile ids, already sorted
    u = depth[0].view(np.uint32).astype(np.uint64)
    dkey = np.where((u & 0x80000000) != 0, (~u) & 0xFFFFFFFF, u | 0x80000000)
    out = vs.copy()
    keys = ks[0]
    bounds = np.flatnonzero(np.diff(keys)) + 1
    for a, b in zip(np.r_[0, bounds], np.r_[bounds, len(keys)]):
        if keys[a] == 0:
            continue
        ids = vs[0, a:b]
        assert np.all(np.diff(ids) > 0)                              # ascending ids inside a tile: what the stable passes rely on
        out[0, a:b] = ids[np.lexsort((ids, dkey[ids]))]
    np.testing.assert_array_equal(out, vs_ref)
    assert int((np.diff(np.sort(dkey[np.unique(vs_ref[0])])) == 0).sum()) > 0     # ties really occurred among emitted splats
