"""Developer A/B tool: step time of the bench workload with the tile scatter (lg_fused_set_option key 2 = 1) and with the stable tile radix
sort it replaces (= 0), on the fresh cloud and after a soak.  Learning rates zeroed while measuring.  usage: python tools/scatter_ab.py [config] [soak]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer
from litegs_amd._lib import lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
soak = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
L = lib()
for i in range(16):
    tr.step(i % 8)
torch.cuda.synchronize()


def measure(label, steps=48):
    for i in range(16):
        tr.step(i % 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(i % 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    for i in range(steps):
        tr.forward_only(i % 8)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:44s} step {dt:7.4f} ms   forward only {df:7.4f} ms   re-runs so far {tr.renderer.fallbacks}", flush=True)


def ab(state):
    lrs = [g["lr"] for g in tr.opt.param_groups]
    step_fn = tr.sched.step
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched.step = lambda: None
    for v, name in ((1, "tile scatter"), (0, "tile radix sort"), (1, "tile scatter"), (0, "tile radix sort")):
        L.lg_fused_set_option(2, v)
        measure(f"{state}: {name}")
    L.lg_fused_set_option(2, 1)
    tr.sched.step = step_fn
    for g, lr in zip(tr.opt.param_groups, lrs):
        g["lr"] = lr


ab("fresh cloud")
for i in range(soak):
    tr.step(i % 8)
torch.cuda.synchronize()
ab(f"after {soak} steps")
