#!/usr/bin/env python
"""Developer tool: how full is the machine over a blend launch?  (Is there a tail, and would splitting long tiles recover it?)

Builds bench.py's training_state (or loads nothing else), turns on lg_debug_wave_clock, runs steps and reads, per wave of the lean
blend forward and the fast blend backward, {start, end, list length, tile, hardware id}.  Prints per launch: its span, the sum of the
wave lifetimes (-> mean resident waves), and the share of the launch x SIMD area spent with 0, 1, 2, 3-4, 5-8 resident waves -- a SIMD
needs ~3 waves to keep its vector pipe busy (a lone wave runs the backward at ~385 ns per splat against ~130 ns per splat and SIMD at
eight waves), so the area at <= 2 waves is what a finer work split could recover, weighted by the lost rate.

    python tools/wave_clock.py [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import build_training_state
from litegs_amd import synthetic as S
from litegs_amd._lib import check, lib


def analyse(name, rec):
    rec = rec[rec[:, 1] > 0]
    t0, t1 = rec[:, 0].astype(np.float64) * 1e-2, rec[:, 1].astype(np.float64) * 1e-2          # us
    n = (rec[:, 2] >> 32).astype(np.int64)
    hw = rec[:, 3] & 0xffffffff
    xcc = (rec[:, 3] >> 32) & 0xf
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    sid = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    begin, end = t0.min(), t1.max()
    span = end - begin
    life = (t1 - t0)
    print(f"{name}: {len(rec)} waves on {len(np.unique(sid))} SIMDs, launch span {span:.1f} us, wave lifetime mean {life.mean():.1f} / max {life.max():.1f} us, "
          f"mean resident waves {life.sum() / span:.0f}, list entries {int(n.sum())}")
    # occupancy histogram: area (SIMD x time) by number of resident waves
    area = np.zeros(10)
    per_rate = []
    for s in np.unique(sid):
        m = sid == s
        ev = np.concatenate([np.stack([t0[m], np.ones(m.sum())], 1), np.stack([t1[m], -np.ones(m.sum())], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        cur, prev = 0, begin
        for t, d in ev:
            area[min(cur, 9)] += t - prev
            prev = t
            cur += int(d)
        area[0] += end - prev
    tot = area.sum()
    print("    share of (SIMD x launch) area by resident waves: " + "  ".join(f"{k}: {100 * area[k] / tot:.1f} %" for k in range(9)))
    # when did the last wave START, and how much work was left then
    last_start = t0.max()
    print(f"    last wave started at {last_start - begin:.1f} us ({100 * (last_start - begin) / span:.0f} % of the span); "
          f"from then on the machine only drains: {span - (last_start - begin):.1f} us")
    # time at which resident waves over the whole chip fall below 50 % / 25 % of the slots in use at the peak
    ev = np.concatenate([np.stack([t0, np.ones(len(t0))], 1), np.stack([t1, -np.ones(len(t1))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    peak = occ.max()
    for frac in (0.75, 0.5, 0.25):
        below = np.nonzero(occ >= frac * peak)[0][-1]
        print(f"    chip-wide resident waves stay >= {int(frac * 100)} % of their peak ({int(peak)}) until {ev[below, 0] - begin:.1f} us ({100 * (ev[below, 0] - begin) / span:.0f} % of the span)")


def by_rank(name, rec, walked_of_tile=None):
    """Slots in schedule order, 16 equal groups: when they start, how long they live, what they walk -- is the order heavy-first in TIME,
    and is a wave's lifetime proportional to its entries?"""
    idx = np.nonzero(rec[:, 1] > 0)[0]
    r = rec[idx]
    begin = r[:, 0].min()
    t0, t1 = (r[:, 0] - begin) * 1e-2, (r[:, 1] - begin) * 1e-2
    tile = (r[:, 2] & 0xffffffff).astype(np.int64)
    w = (r[:, 2] >> 32).astype(np.int64) if walked_of_tile is None else walked_of_tile[tile]
    print(f"    {name}: slots in schedule order, 16 groups: entries walked (mean / max), start (mean), lifetime (mean / max), end (mean / max), ns per entry")
    for g in np.array_split(np.arange(len(idx)), 16):
        print(f"      slots {idx[g[0]]:6d}..{idx[g[-1]]:6d}: walked {w[g].mean():7.1f} / {w[g].max():5d}   start {t0[g].mean():7.1f}   life {(t1 - t0)[g].mean():6.1f} / {(t1 - t0)[g].max():6.1f}"
              f"   end {t1[g].mean():7.1f} / {t1[g].max():7.1f}   {1e3 * (t1 - t0)[g].sum() / max(w[g].sum(), 1):6.1f}")
    walked = np.zeros(int(tile.max()) + 1, dtype=np.int64)
    walked[tile] = w
    return walked


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n, W, H, f = S.CONFIGS["3m_1080p"]
    tr, _ = build_training_state(n, W, H, f, S.make_scene(n, seed=0), 8, 60)
    for k in range(8):
        tr.step(k)
    tr.flush()
    torch.cuda.synchronize()
    slots = 16384
    fwd = torch.zeros((slots, 4), dtype=torch.int64, device="cuda")
    bwd = torch.zeros((slots, 4), dtype=torch.int64, device="cuda")
    L = lib()
    check(L.lg_debug_wave_clock(fwd.data_ptr(), bwd.data_ptr()), "wave clock")
    try:
        for i in range(steps):
            fwd.zero_(); bwd.zero_()
            tr.step(i % 8)
            tr.flush()
            torch.cuda.synchronize()
            print(f"--- step {i} (camera {i % 8})")
            f_rec, b_rec = fwd.cpu().numpy(), bwd.cpu().numpy()
            analyse("blend forward (lean)", f_rec)
            analyse("blend backward (fast)", b_rec)
            walked = by_rank("backward", b_rec)
            by_rank("forward", f_rec, walked)
    finally:
        check(L.lg_debug_wave_clock(None, None), "wave clock off")


if __name__ == "__main__":
    main()
