"""Developer A/B tool: in-situ blend-backward time of the bench workload, generic kernel vs the 8x16 fast kernel vs the splat-parallel variant (lg_set_tuning key 5: 0 / 1 / 2),
on the fresh cloud and after a soak of training steps (the trained-state cloud).  Learning rates are zeroed while measuring so that both
variants see the same cloud.  usage: python tools/bwd_ab.py [config] [soak_steps]"""
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


def measure(label, steps=32):
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
    print(f"{label:46s} step {dt:7.4f} ms   blend backward avg {sum(ts) / len(ts):7.4f} min {ts[0]:7.4f} max {ts[-1]:7.4f}", flush=True)


def ab(state):
    lrs = [g["lr"] for g in tr.opt.param_groups]
    step_fn = tr.sched.step
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched.step = lambda: None
    for v, name in ((0, "generic"), (1, "fast"), (2, "splat-parallel"), (1, "fast"), (2, "splat-parallel")):
        L.lg_set_tuning(5, v)
        measure(f"{state}: {name} kernel")
    L.lg_set_tuning(5, 1)
    for v, name in ((0, "generic"), (1, "packed"), (0, "generic"), (1, "packed")):      # blend forward: generic loop vs the packed 8x16 loop
        L.lg_set_tuning(7, v)
        for i in range(8):
            tr.forward_only(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(64):
            tr.forward_only(i % 8)
        torch.cuda.synchronize()
        print(f"{state}: forward only, {name} blend forward   {(time.perf_counter() - t0) / 64 * 1e3:7.4f} ms", flush=True)
    L.lg_set_tuning(7, 1)
    L.lg_set_tuning(4, 0)
    measure(f"{state}: fast kernel, no tile schedule")
    L.lg_set_tuning(4, 1)
    tr.sched.step = step_fn
    for g, lr in zip(tr.opt.param_groups, lrs):
        g["lr"] = lr


ab("fresh cloud")
for i in range(soak):
    tr.step(i % 8)
torch.cuda.synchronize()
ab(f"after {soak} steps")
