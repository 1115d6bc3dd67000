"""Developer tool: distribution of the tile lists' lengths the executor builds -- unculled first visit and culled steady state -- at the
bench workload.  usage: python tools/tile_list_stats.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from litegs_amd import synthetic as S
from litegs_amd._lib import lib
from litegs_amd.trainer import SyntheticTrainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "3m_1080p"
n, W, H, f = S.CONFIGS[cfg]
tr = SyntheticTrainer(n, W, H, f, n_frames=8)
R = tr.renderer


def lengths():
    torch.cuda.synchronize()
    ws2, L, N = R.last_ws2
    off = lib().lg_fused_tile_start_offset(L, N, R.H, R.W, R.TH, R.TW)
    st = ws2[off:off + 4 * (R.ntiles + 2)].view(torch.int32).cpu().numpy()
    a, b = st[1:R.ntiles + 1], st[2:R.ntiles + 2]
    return np.where((a >= 0) & (b > a), b - a, 0)


def report(tag, n_):
    q = np.percentile(n_, [50, 90, 99, 99.9])
    print(f"{tag:34s} tiles {len(n_)}  sum {int(n_.sum())}  mean {n_.mean():.0f}  p50 {q[0]:.0f}  p90 {q[1]:.0f}  p99 {q[2]:.0f}  p99.9 {q[3]:.0f}  max {n_.max()}  "
          f">512: {int((n_ > 512).sum())}  >1024: {int((n_ > 1024).sum())}  >2048: {int((n_ > 2048).sum())}", flush=True)


with torch.no_grad():
    tr.forward_only(0)
report("first visit (unculled)", lengths())
for i in range(24):
    tr.step(i % 8)
with torch.no_grad():
    tr.forward_only(0)
report(f"steady state (culled={R.last_cull})", lengths())
