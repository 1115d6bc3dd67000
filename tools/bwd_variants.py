"""Developer A/B tool: step time and in-situ blend-backward time of the 3M bench workload for launch variants (lg_set_tuning).
Learning rates are zeroed so the cloud (and with it the workload) stays what it was at the start."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer
from litegs_amd._lib import lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
for g in tr.opt.param_groups:
    g["lr"] = 0.0
tr.sched.step = lambda: None
L = lib()
for i in range(16):
    tr.step(i % 8)
torch.cuda.synchronize()


def measure(label, steps=16):
    for i in range(8):
        tr.step(i % 8)
    torch.cuda.synchronize()
    ev = []
    tr.renderer.probe_events = ev
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(i % 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    tr.renderer.probe_events = None
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{label:40s} step {dt:7.4f} ms   blend backward avg {sum(ts) / len(ts):7.4f} min {ts[0]:7.4f} max {ts[-1]:7.4f}", flush=True)


measure("baseline (heaviest-first schedule)")
L.lg_set_tuning(4, 0)
measure("no schedule (band per XCD)")
L.lg_set_tuning(4, 1)
measure("schedule again")
