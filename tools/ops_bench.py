"""Developer tool: step time of the operator-by-operator path (the litegs_fused drop-in surface) next to the fused executor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
for fused in (False, True):
    tr = SyntheticTrainer(n, W, H, f, n_frames=8, fused=fused)
    for i in range(24):
        tr.step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    K = 40
    for i in range(K):
        tr.step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"{cfg} fused={fused}: {dt*1e3:.3f} ms/step  {1/dt:.1f} frames/s", flush=True)
    del tr
    torch.cuda.empty_cache()
