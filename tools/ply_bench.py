#!/usr/bin/env python
"""Developer tool: wall time of save_ply / load_ply (litegs_amd/io/ply.py) at N Gaussians, next to the reference's construction
(`elements[:] = list(map(tuple, attributes))`, litegs/io_manager/ply.py:40-42) timed on a sample and extrapolated."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litegs_amd.io import ply

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
rng = np.random.default_rng(0)
t = [rng.standard_normal(s, dtype=np.float32) for s in ((3, N), (3, N), (4, N), (1, 3, N), (15, 3, N), (1, N))]
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "pc.ply")
    t0 = time.time(); ply.save_ply(p, *t); t1 = time.time()
    size = os.path.getsize(p)
    back = ply.load_ply(p, 3); t2 = time.time()
    assert all(np.array_equal(a, b) for a, b in zip(t, back))
print(f"N={N}: save {t1 - t0:.2f} s ({size / 1e6:.0f} MB, {size / 1e6 / (t1 - t0):.0f} MB/s), load {t2 - t1:.2f} s")
M = 100_000
attrs = rng.standard_normal((M, 62)).astype(np.float32)
dt = [(f"a{i}", "f4") for i in range(62)]
t0 = time.time(); el = np.empty(M, dtype=dt); el[:] = list(map(tuple, attrs)); t1 = time.time()
print(f"reference-style tuple construction: {t1 - t0:.2f} s per {M} points -> {(t1 - t0) * N / M:.0f} s at N={N}")
