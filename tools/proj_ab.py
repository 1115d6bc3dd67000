"""Developer A/B tool: the fused projection with the SH loads issued in front of the tile walk (lg_set_tuning(12, 1), csrc/fused.hip
project_fused_kernel<.., EARLY>) against the default (SH loads behind the walk, only for splats that are emitted), in situ on the bench
workload: fresh cloud and after a soak.  Learning rates are zeroed while measuring; the variants alternate so that drift cancels.
usage: python tools/proj_ab.py [config] [soak_steps]"""
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
tr.speculative = True
L = lib()
for i in range(24):
    tr.step(i % 8)
tr.flush()


def measure(label, steps=64):
    for i in range(8):
        tr.step(i % 8)
    tr.flush()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        tr.step(i % 8)
        ev[i + 1].record()
    tr.flush()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    t0 = time.perf_counter()
    for i in range(steps):
        tr.forward_only(i % 8)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:44s} step p50 {ms[steps // 2]:7.4f} mean {sum(ms) / steps:7.4f} ms   forward only {fwd:7.4f} ms", flush=True)


def ab(state):
    lrs = [g["lr"] for g in tr.opt.param_groups]
    step_fn = tr.sched.step
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched.step = lambda: None
    for v, name in ((0, "SH behind the walk"), (1, "SH in front of the walk")) * 3:
        L.lg_set_tuning(12, v)
        measure(f"{state}: {name}")
    L.lg_set_tuning(12, 0)
    tr.sched.step = step_fn
    for g, lr in zip(tr.opt.param_groups, lrs):
        g["lr"] = lr


ab("fresh cloud")
for i in range(soak):
    tr.step(i % 8)
tr.flush()
ab(f"after {soak} steps")
