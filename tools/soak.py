"""Developer tool: a longer training run of the bench workload -- step time per block of steps, unculled re-runs, bound margins, Gaussians
with Adam history, finiteness of the parameters.  usage: python tools/soak.py [config] [steps] [block]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
block = int(sys.argv[3]) if len(sys.argv) > 3 else 250
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
R = tr.renderer
for i in range(8):
    tr.step(i)
torch.cuda.synchronize()
done, fb = 0, R.fallbacks
while done < steps:
    t0 = time.perf_counter()
    for i in range(block):
        tr.step((done + i) % 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / block * 1e3
    done += block
    finite = all(bool(torch.isfinite(p).all()) for p in tr.params)
    hist = int(tr.fadam.touched.sum()) if tr.fadam.touched is not None else -1
    print(f"steps {done - block:5d}-{done:5d}: {dt:7.4f} ms/step  unculled re-runs {R.fallbacks - fb:3d}  margins {sorted(set(R.margin))}  emitted {int(R.fb_total[0])} "
          f"full {R.full_total[0]}  Gaussians with Adam history {hist}  loss {float(tr.last['loss']):.4f}  finite {finite}", flush=True)
    fb = R.fallbacks
