"""The step real training runs late in a 30 000-iteration density-control run, isolated so that it can be measured and profiled.

tests/convergence_3m.py (LITEGS_CONV_SAVE=<file>:<epoch>) stores the student cloud and its Adam moments at an epoch boundary of the 3 M /
150-camera run; this tool loads it, re-renders the teacher targets of a handful of cameras and replays training steps on it in the regimes
`training.start` is in at that point (reference loop: litegs/training/trainer.py:108-195, render/__init__.py:75-79):

    plain steps along the statistics helper's cached tile list (every render after the first statistics epoch),
    statistics-epoch steps (statistic-mode blend kernels + scatter of the moments),
    plain steps on the executor's own schedule + depth-bound culling (stat_schedule_always = False),

each under the executor's list-building variants (tile scatter + per-tile sort with / without the workgroup radix regime, splat sort +
tile radix sort for long lists).  Learning rates are zero and the Adam moments are restored between variants, so every variant sees the
same cloud.

    python tools/late_phase.py ab    /tmp/late.pt            all variants, ms per step from events (one process)
    python tools/late_phase.py trace /tmp/late.pt [variant]  a few steps of one variant for rocprofv3 --kernel-trace / --pmc
    python tools/late_phase.py parity /tmp/late.pt           executor tables and image of late-phase frames against the oracle (bit-exact tables)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from litegs_amd import densify as D
from litegs_amd import synthetic as S
from litegs_amd.statistics import STATS
from litegs_amd.trainer import SyntheticTrainer

FRAMES = int(os.environ.get("LATE_FRAMES", "16"))           # cameras replayed (of the run's 150, evenly spaced)
N3M, W, H, FOCAL, SEED, RUN_FRAMES = 3_000_000, 1920, 1080, 1200.0, 0, 150

VARIANTS = {
    # name: (renderer attributes, statistics epoch?)
    "default": (dict(), False),
    "wg_radix": (dict(tilesort_wg_radix=True), False),
    "long_global_1000": (dict(long_list_global=1000), False),
    "global": (dict(depth_order=0), False),
    "own_schedule": (dict(stat_schedule_always=False), False),
    "own_schedule_wg_radix": (dict(stat_schedule_always=False, tilesort_wg_radix=True), False),
    "stat_epoch": (dict(), True),
    "stat_epoch_wg_radix": (dict(tilesort_wg_radix=True), True),
}


def load(path):
    st = torch.load(path, map_location="cpu")
    scene = [p.numpy() for p in st["params"]]
    n = scene[0].shape[-2] * scene[0].shape[-1]
    # the run's cameras: orbit_cameras(150, ...); the replay uses every (150 // FRAMES)-th of them
    tr = SyntheticTrainer(n, W, H, FOCAL, n_frames=RUN_FRAMES, seed=SEED, scene=scene, noise_targets=False)
    pick = list(range(0, RUN_FRAMES, RUN_FRAMES // FRAMES))[:FRAMES]
    teacher = SyntheticTrainer(N3M, W, H, FOCAL, n_frames=RUN_FRAMES, seed=SEED, scene=S.make_scene(N3M, seed=SEED), noise_targets=False)
    for k in pick:
        tr.frames[k].gt = teacher.forward_only(k).clamp(0, 1).clone()
    teacher.close()
    del teacher
    torch.cuda.empty_cache()
    tr.degree = int(st["degree"])
    tr.fadam._init_state()
    by = {g["name"]: g["params"][0] for g in tr.opt.param_groups}
    order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
    moments = [(st["exp_avg"][i].cuda(), st["exp_avg_sq"][i].cuda()) for i in range(6)]

    def restore():
        for i, nme in enumerate(order):
            tr.opt.state[by[nme]]["exp_avg"].copy_(moments[i][0])
            tr.opt.state[by[nme]]["exp_avg_sq"].copy_(moments[i][1])
        tr.fadam.touched = None
    restore()
    for g in tr.opt.param_groups:
        g["lr"] = 0.0                                       # every variant sees the same cloud
    tr.sched.step = lambda: None
    tr.speculative = True
    tr.enable_densify(D.DensifyParams(target_primitives=int(1.1 * N3M)), total_epochs=200, seed=SEED)
    return tr, pick, restore, st


def stat_pass(tr, pick):
    """one statistics epoch over the replayed cameras: leaves the cached heavy-first tile list of every frame, as in the run"""
    STATS.active = True
    try:
        for k in pick:
            tr.step(k)
    finally:
        STATS.active = False
    tr.flush()


def configure(tr, attrs):
    rd = tr.renderer
    base = dict(tilesort_wg_radix=False, long_list_global=0, depth_order=2, stat_schedule_always=True)
    base.update(attrs)
    for k, v in base.items():
        setattr(rd, k, v)
    tr.flush()
    rd.reset_feedback()


def timed(tr, pick, rounds, stat):
    """ms per step over rounds x len(pick) steps (events on the launch stream), after one untimed round"""
    def one_round():
        if stat:
            STATS.active = True
        try:
            for k in pick:
                tr.step(k)
        finally:
            STATS.active = False
    one_round(); one_round()
    tr.flush()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        one_round()
    tr.flush()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (rounds * len(pick))


def forward_ms(tr, pick, rounds):
    with torch.no_grad():
        for k in pick:
            tr.forward_only(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            for k in pick:
                tr.forward_only(k)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (rounds * len(pick)) * 1e3


def list_lengths(tr):
    import ctypes
    from litegs_amd._lib import lib
    rd = tr.renderer
    torch.cuda.synchronize()
    ws2, L, N = rd.last_ws2
    off = lib().lg_fused_tile_start_offset(L, N, rd.H, rd.W, rd.TH, rd.TW)
    st = ws2[off:off + 4 * (rd.ntiles + 2)].view(torch.int32).cpu().numpy()
    a, b = st[1:rd.ntiles + 1], st[2:rd.ntiles + 2]
    return np.where((a >= 0) & (b > a), b - a, 0)


def main():
    mode, path = sys.argv[1], sys.argv[2]
    tr, pick, restore, st = load(path)
    rd = tr.renderer
    print(f"late-phase cloud: epoch {st['epoch']}, {tr.n_chunks * tr.S} Gaussians, SH degree {tr.degree}, {len(pick)} of {RUN_FRAMES} cameras", flush=True)
    stat_pass(tr, pick)                                     # the run had its statistics epochs: every frame has a cached tile list
    if mode == "ab":
        with torch.no_grad():
            tr.forward_only(pick[0])
        n_ = list_lengths(tr)
        q = np.percentile(n_, [50, 90, 99])
        print(f"tile lists of camera {pick[0]}: instances {int(n_.sum())}  mean {n_.mean():.0f}  p50 {q[0]:.0f}  p90 {q[1]:.0f}  p99 {q[2]:.0f}  max {n_.max()}  "
              f"tiles > 1024: {int((n_ > 1024).sum())}  > 2048: {int((n_ > 2048).sum())}  > 4096: {int((n_ > 4096).sum())}", flush=True)
        names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(VARIANTS)
        for name in names:
            attrs, stat = VARIANTS[name]
            restore()
            configure(tr, attrs)
            fb0, rp0 = rd.fallbacks, tr.spec_replays
            ms = timed(tr, pick, 3, stat)
            fw = forward_ms(tr, pick, 2)
            inst = float(np.mean([rd.fb_total[k] for k in pick]))
            print(f"{name:24s} step {ms:7.3f} ms   forward only {fw:7.3f} ms   emitted instances / frame {inst / 1e6:6.2f} M   culled last: {rd.last_cull}   "
                  f"unculled re-runs {rd.fallbacks - fb0}  replayed steps {tr.spec_replays - rp0}", flush=True)
    elif mode == "trace":
        name = sys.argv[3] if len(sys.argv) > 3 else "default"
        attrs, stat = VARIANTS[name]
        restore()
        configure(tr, attrs)
        steps = int(os.environ.get("LATE_TRACE_STEPS", "8"))
        for k in pick:                                      # first visits (blocking sizes) stay outside the traced window's tail
            tr.step(k)
        tr.flush()
        print("TRACE_BEGIN", flush=True)
        if stat:
            STATS.active = True
        try:
            for i in range(steps):
                tr.step(pick[i % len(pick)])
        finally:
            STATS.active = False
        tr.flush()
        torch.cuda.synchronize()
        print(f"traced {steps} steps of variant {name}; emitted instances / frame {np.mean([rd.fb_total[k] for k in pick]) / 1e6:.2f} M", flush=True)
    elif mode == "parity":
        import ctypes
        from litegs_amd._lib import lib
        from oracle import oracle as O
        L = lib()
        scene = [p.detach().cpu().numpy() for p in tr.params]
        bad = 0
        for attrs_name in ("default", "wg_radix", "global"):
            configure(tr, VARIANTS[attrs_name][0])
            rd.stat_schedule_always = False                 # the executor's own path end to end (tables are the same either way)
            for k in pick[:int(os.environ.get("LATE_PARITY_FRAMES", "3"))]:
                fr = tr.frames[k]
                with torch.no_grad():
                    img = tr.forward_only(k)
                torch.cuda.synchronize()
                ref = O.render_forward(scene, fr.view.cpu().numpy(), fr.proj.cpu().numpy(), fr.planes.cpu().numpy(), H, W, tr.degree)
                ws2, tl, N = rd.last_ws2
                o_ts = L.lg_fused_tile_start_offset(tl, N, H, W, 8, 16)
                o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), tl, N, H, W, 8, 16)
                ts = ws2[o_ts:o_ts + 4 * (rd.ntiles + 2)].view(torch.int32).cpu().numpy()
                total = int(rd.fb_total[k])
                ok_n = abs(total - ref.n_instances) <= max(2, int(2e-6 * ref.n_instances))
                same_ts = total == ref.n_instances and np.array_equal(ts, ref.tile_start[0])
                pts = ws2[o_pts:o_pts + 4 * total].view(torch.int32).cpu().numpy()
                # splat ids are positions in the compacted arrays; the oracle compacts the same visible chunks in the same (ascending) order
                same_pts = same_ts and np.array_equal(pts, ref.sorted_point[0][:total])
                err = np.abs(img.cpu().numpy() - np.clip(ref.img[..., :H, :W], 0, 1))
                flips = int((err > 1e-4).sum())
                print(f"parity {attrs_name:10s} camera {k:3d}: instances {total} (oracle {ref.n_instances})  count ok {ok_n}  tile ranges identical {same_ts}  "
                      f"lists identical {same_pts}  image max |d| {err.max():.2e}  pixels beyond 1e-4: {flips} of {err.size}", flush=True)
                bad += 0 if (ok_n and (same_pts or total != ref.n_instances) and flips <= 250) else 1
        print("LATE_PHASE_PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
    tr.close()


if __name__ == "__main__":
    main()
