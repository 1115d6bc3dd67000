"""Developer tool: time lg_radix_sort_pairs alone (tile-sort-like and depth-sort-like inputs) against a plain device copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import fused as F

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
for n, bits, name in ((11_700_000, 14, "tile"), (1_087_000, 32, "depth")):
    if bits == 14:
        # tile-like keys: runs of ~20 neighbouring tiles per splat
        base = torch.randint(0, 8100, (n // 16 + 1,), device="cuda", generator=g)
        keys = (base[:, None] + torch.arange(16, device="cuda")[None, :] % 4 + (torch.arange(16, device="cuda")[None, :] // 4) * 120).reshape(-1)[:n].clamp(0, 8100).to(torch.int32)
    else:
        keys = torch.randint(0, 2**31 - 1, (n,), device="cuda", generator=g, dtype=torch.int32)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ref = torch.sort(keys.to(torch.int64), stable=True)
    dst_k, dst_v = torch.empty_like(keys), torch.empty_like(vals)
    def fresh_sort():            # the sort destroys its inputs (ping-pong): give it a fresh copy every time
        dst_k.copy_(keys); dst_v.copy_(vals)
        return F.radix_sort_pairs(dst_k, dst_v, 0, bits)
    k2, v2 = fresh_sort()
    torch.cuda.synchronize()
    ok = torch.equal(k2.to(torch.int64), ref.values) and torch.equal(v2.to(torch.int64), ref.indices)
    tc = timeit(lambda: (dst_k.copy_(keys), dst_v.copy_(vals)))
    t = timeit(fresh_sort) - tc
    passes = (bits + 7) // 8
    print(f"{name}: n={n} bits={bits} correct={ok} sort {t:.1f} us ({t/passes:.1f} us/pass incl. setup) ; copy of the pairs {tc:.1f} us", flush=True)
