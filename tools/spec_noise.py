"""Developer tool: how far apart do the parameters of the speculative-culling test scenario end up -- gated repeat vs gated repeat (the
atomics-order noise floor), speculative vs speculative, gated vs speculative -- per parameter tensor, a few repetitions.  Calibrates the
tolerance of tests/test_gpu_cull.py::test_speculative_culling_replays_failed_steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from litegs_amd.trainer import SyntheticTrainer

NAMES = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]


def run(speculative, sabotage=True, drop_step=-1):
    tr = SyntheticTrainer(150_000, 640, 360, 380.0, n_frames=2, seed=5)
    tr.speculative = speculative
    rd = tr.renderer
    losses = []
    for i in range(14):
        if sabotage and i in (6, 9):
            torch.cuda.synchronize()
            k = i % 2
            gx, gy = -(-640 // 16), -(-360 // 8)
            upper = sum((-(-gx // (1 << q))) * (-(-gy // (1 << q))) for q in range(1, 4))
            rd.sched[k, rd.sched_cur[k]][:upper + gx * gy].view(torch.float32).mul_(0.2)
        if i == drop_step:                                   # what the test has to detect: one step's update is missing
            continue
        tr.step(i % 2)
        losses.append(tr.last["loss"])
    tr.flush()
    torch.cuda.synchronize()
    out = ([p.detach().clone() for p in tr.params], [float(l) for l in losses], tr.renderer.fallbacks, tr.spec_replays)
    tr.close()
    return out


def dist(a, b):
    return [float((x - y).abs().mean()) for x, y in zip(a[0], b[0])]          # mean |difference| (the max is dominated by single sign flips)


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    g1, g2, s1, s2 = run(False), run(False), run(True), run(True)
    n1, n2 = run(False, False), run(True, False)
    print(f"rep {rep}: fallbacks gated {g1[2]}/{g2[2]}  replays spec {s1[3]}/{s2[3]}")
    for tag, d in (("gated - gated", dist(g1, g2)), ("spec  - spec ", dist(s1, s2)), ("gated - spec ", dist(g1, s1)), ("gated - spec2", dist(g2, s2)),
                   ("no sabotage: gated - spec", dist(n1, n2)), ("sabotage vs none (gated)", dist(g1, n1)),
                   ("gated - gated with step 12 lost", dist(g1, run(False, True, 12)))):
        print(f"   {tag:28s} " + "  ".join(f"{n} {x:.2e}" for n, x in zip(NAMES, d)))
    print("   losses gated", " ".join(f"{x:.5f}" for x in g1[1]))
    print("   losses spec ", " ".join(f"{x:.5f}" for x in s1[1]), flush=True)
