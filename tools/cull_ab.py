"""Developer A/B tool: step time of the 3M bench workload with the depth-bound culling on / off (frozen parameters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
for g in tr.opt.param_groups:
    g["lr"] = 0.0
tr.sched.step = lambda: None


def measure(label, steps=32):
    for i in range(16):
        tr.step(i % 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(i % 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:40s} step {dt:7.4f} ms  emitted instances {int(tr.renderer.fb_total[0])}  full {tr.renderer.full_total[0]}", flush=True)


for rep in range(2):
    tr.renderer.cull_enabled = True
    measure("depth-bound culling ON")
    tr.renderer.cull_enabled = False
    measure("depth-bound culling OFF")
