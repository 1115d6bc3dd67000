"""Quick stage timing probe (developer tool): python tools/probe.py [config] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "500k_1080p"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, W, H, f = S.CONFIGS[cfg]
t0 = time.time()
tr = SyntheticTrainer(n, W, H, f, n_frames=8, loss_fn=None)
print("setup s", time.time() - t0, flush=True)
for i in range(16):
    tr.step(i)
torch.cuda.synchronize()
print("stats", tr.workload_stats(0), flush=True)
t0 = time.time()
for i in range(steps):
    tr.step(i)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"train step {dt*1e3:.3f} ms  -> {1/dt:.1f} it/s", flush=True)
t0 = time.time()
for i in range(steps):
    tr.forward_only(i)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"forward {dt*1e3:.3f} ms -> {n/dt/1e6:.1f} Msplats/s", flush=True)
# host-only overhead estimate: time to enqueue
t0 = time.time()
for i in range(steps):
    tr.step(i)
t_enq = (time.time() - t0) / steps
torch.cuda.synchronize()
print(f"host enqueue per step {t_enq*1e3:.3f} ms", flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(8):
        tr.step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
