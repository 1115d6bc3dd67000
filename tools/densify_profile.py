"""Developer tool: wall time per epoch of a density-control training run at the bench size (3 M Gaussians @1080p), executor or operator path.
usage: python tools/densify_profile.py executor|operator [frames] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S, densify as D
from litegs_amd.trainer import SyntheticTrainer

fused = sys.argv[1] == "executor"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n, W, H, f = S.CONFIGS["3m_1080p"]
tr = SyntheticTrainer(n, W, H, f, n_frames=frames, fused=fused)
ctl = tr.enable_densify(D.DensifyParams(target_primitives=int(1.1 * n)), total_epochs=200, seed=0)
T0 = time.perf_counter()
for epoch in range(epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.degree = min(epoch // 5, 3)
    with tr.begin_epoch(epoch):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for k in range(frames):
            tr.step(k)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    tr.end_epoch(epoch)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{sys.argv[1]} epoch {epoch:3d}: begin {1e3 * (t1 - t0):8.1f} ms   steps {1e3 * (t2 - t1) / frames:8.3f} ms/step   end {1e3 * (t3 - t2):8.1f} ms   "
          f"points {tr.n_chunks * tr.S}  re-runs {tr.renderer.fallbacks}  t = {time.perf_counter() - T0:7.1f} s", flush=True)
