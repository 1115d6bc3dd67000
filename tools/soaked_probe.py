"""Developer tool: the trained-state cloud of the bench workload as a file, so that a profiler (rocprofv3 --pmc serialises every dispatch:
a 1000-step soak under it takes many minutes) only has to wrap a few steps.
    python tools/soaked_probe.py save /tmp/soaked.npz [soak_steps]     train the bench workload, store the raw parameters
    python tools/soaked_probe.py run  /tmp/soaked.npz [steps]           8 first visits + `steps` training steps on the stored cloud (lr 0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

mode, path = sys.argv[1], sys.argv[2]
n, W, H, f = S.CONFIGS["3m_1080p"]
if mode == "save":
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    tr = SyntheticTrainer(n, W, H, f, n_frames=8)
    for i in range(8 + steps):
        tr.step(i % 8)
    torch.cuda.synchronize()
    np.savez(path, *[p.detach().cpu().numpy() for p in tr.params])
    print("saved", path, "Gaussians with Adam history", int(tr.fadam.touched.sum()))
else:
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    z = np.load(path)
    scene = [z[f"arr_{k}"] for k in range(6)]
    tr = SyntheticTrainer(n, W, H, f, n_frames=8, scene=scene)
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched.step = lambda: None
    for i in range(8 + steps):
        tr.step(i % 8)
    torch.cuda.synchronize()
    print("ran", steps, "steps; emitted", int(tr.renderer.fb_total[0]), "full", tr.renderer.full_total[0])
