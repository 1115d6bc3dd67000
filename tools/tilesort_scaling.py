#!/usr/bin/env python
"""Developer tool: the per-tile depth sort as a function of the list length (16 200 tiles, every list the same length, random depths):
how the one-wave radix regime (<= 1024), the workgroup bitonic regime (<= 2048) and the chunked regime beyond it scale.
usage: python tools/tilesort_scaling.py [lengths ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from litegs_amd._lib import lib, check

L_ = lib()
ntiles, N = 16200, 1_200_000
lengths = [int(x) for x in sys.argv[1:]] or [250, 700, 1000, 1500, 2500, 4000]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
depth = torch.rand((N,), device=dev, generator=g) * 20 + 0.5
s = torch.cuda.current_stream().cuda_stream
for n in lengths:
    total = ntiles * n
    ts = torch.full((ntiles + 2,), -1, dtype=torch.int32, device=dev)
    ts[1:ntiles + 2] = torch.arange(0, ntiles + 1, device=dev, dtype=torch.int32) * n
    base = torch.randint(0, N, (total,), device=dev, generator=g, dtype=torch.int32)
    scratch = torch.empty((total,), dtype=torch.int32, device=dev)
    ms = []
    for rep in range(4):
        vals = base.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        check(L_.lg_tile_depth_sort_unordered(vals.data_ptr(), ts.data_ptr(), depth.data_ptr(), 1, total, N, ntiles, scratch.data_ptr(), s), "sort")
        b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    d = depth[vals[: n].long()]
    ok = bool((d[1:] >= d[:-1]).all())
    print(f"list length {n:5d}: {total / 1e6:6.1f} M instances  {min(ms[1:]):8.3f} ms  ({min(ms[1:]) * 1e3 / (total / 1e6):7.1f} us per M instances)  first tile sorted: {ok}", flush=True)
