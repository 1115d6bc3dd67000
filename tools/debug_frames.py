"""Debug tool: render many orbit frames of the 3 M scene once each (first visits), printing the frame index before every render.
usage: [HIP_LAUNCH_BLOCKING=1] python -X faulthandler tools/debug_frames.py [frames] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd.trainer import SyntheticTrainer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
tr = SyntheticTrainer(n, 1920, 1080, 1200.0, n_frames=frames, noise_targets=False)
for k in range(frames):
    print("frame", k, flush=True)
    img = tr.forward_only(k)
    torch.cuda.synchronize()
    print("   ok: visible chunks", int(tr.renderer.fb_vis[k]), "instances", int(tr.renderer.fb_total[k]), "table", tr.renderer.last_sizes, flush=True)
print("all frames rendered")
