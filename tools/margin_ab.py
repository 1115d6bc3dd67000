"""Developer A/B tool: step time of the bench workload (training, parameters move) for depth-order mode x bound margin, with the
number of visits that fell back to the unculled run.  usage: python tools/margin_ab.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litegs_amd import synthetic as S
from litegs_amd._lib import lib
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n, W, H, f = S.CONFIGS[cfg]
scene = S.make_scene(n, seed=0)
L = lib()
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ("global", "tile")
margins = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (0, 50, 100, 200)
for mode in modes:
    for margin in margins:                    # 0 = adaptive
        L.lg_fused_set_option(0, 1 if mode == "tile" else 0)
        tr = SyntheticTrainer(n, W, H, f, n_frames=8, scene=scene)
        R = tr.renderer
        R.margin_fixed = margin
        R.reset_feedback()
        for i in range(8 + 16):
            tr.step(i % 8)
        torch.cuda.synchronize()
        fb0 = R.fallbacks
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(i % 8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        fb1 = R.fallbacks
        for i in range(8):                     # fallbacks are observed one visit late: flush
            tr.step(i % 8)
        torch.cuda.synchronize()
        print(f"{mode:6s} margin {('adaptive' if margin == 0 else str(margin)):8s} step {dt:7.4f} ms  fallbacks in window ~{R.fallbacks - fb0 - (R.fallbacks - fb1) + 0:3d} (+{R.fallbacks - fb1} late)  "
              f"emitted {int(R.fb_total[0])} full {R.full_total[0]}  margins {R.margin}", flush=True)
        del tr
L.lg_fused_set_option(0, 2)
