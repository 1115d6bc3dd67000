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
    "default": (dict(), False),                                            # whatever FusedRenderer defaults to
    "tile": (dict(long_list_global=0), False),                             # tile scatter + per-tile depth sort whatever the list lengths
    "global": (dict(depth_order=0), False),                                # splat sort + stable tile radix sort
    "own_schedule": (dict(stat_schedule_always=False), False),             # default list building, executor's schedule + depth-bound culling
    "own_schedule_tile": (dict(stat_schedule_always=False, long_list_global=0), False),
    "no_replicas": (dict(replicas_enabled=False), False),                  # gradient replicas off (blend backward contends, nothing to fold)
    "sched_refresh": (dict(refresh_stat_schedule=True), False),
    "sort_packed_keys": (dict(_tuning={26: 1}), False),                    # packed between the passes, keys written by the second pass + tile_range (first packed form)
    "sort_unpacked": (dict(_tuning={26: 0}), False),                       # tile radix sort with separate key / value arrays between its two passes (rounds 2-5)
    "seg": (dict(_tuning={22: 1}), False),                                 # segmented blend backward (checkpoints every 512 list positions)
    "seg256": (dict(_tuning={22: 1, 23: 8}), False),                       # segments of 256 / 1024 / 128 list positions
    "seg1024": (dict(_tuning={22: 1, 23: 10}), False),
    "seg128": (dict(_tuning={22: 1, 23: 7}), False),
    "lean_s96": (dict(_tuning={17: 2}), False),                            # lean blend forward without the 80-register cap (7 waves per SIMD under the trap handler)
    "lean_s96_sched_refresh": (dict(refresh_stat_schedule=True, _tuning={17: 2}), False),            # the helper's tile list re-ordered after every render, not only in statistics epochs
    "prio": (dict(_tuning={8: 1}), False),                                 # issue priority by rank in the heavy-first schedule (csrc/raster.hip wave_rank_priority)
    "prio_stat_epoch": (dict(_tuning={8: 1}), True),
    "global_prio": (dict(depth_order=0, _tuning={8: 1}), False),
    "emit_dynamic": (dict(_tuning={11: 2}), False),                        # key emission: groups handed out on demand after one static round (default: round robin)
    "emit_hi128": (dict(_tuning={10: 128}), False),                        # ... in-workgroup walk up to 128 tiles (default 256), larger splats cooperative
    "emit_hi64": (dict(_tuning={10: 64}), False),
    "sort_lb32": (dict(_tuning={15: 32}), False),                          # 32-wide look-back in the splat sort
    "occ6": (dict(_tuning={19: 26, 20: 26}), False),                      # blend kernels capped at 6 / 5 / 4 / 3 waves per SIMD (unused dynamic LDS)
    "occ5": (dict(_tuning={19: 32, 20: 32}), False),
    "occ4": (dict(_tuning={19: 40, 20: 40}), False),
    "occ3": (dict(_tuning={19: 53, 20: 53}), False),
    "occ4_bwd": (dict(_tuning={20: 40}), False),
    "occ4_fwd": (dict(_tuning={19: 40}), False),
    "probe_fwd_cached": (dict(_tuning={18: 2}), False),                   # measurement hooks (wrong results): forward reads 1024 always-cached records
    "probe_bwd_cached": (dict(_tuning={18: 5}), False),                   # backward: the same, and no atomics
    "probe_both_cached": (dict(_tuning={18: 7}), False),
                            # blend launches: one wave per tile, heaviest first (round 5) instead of the snake schedule
    "dhist256": (dict(_tuning={24: 256}), False) ,                         # depth_keys_hist_kernel: 256 workgroups (rounds 3-5; default 1024)
    "lean0": (dict(_tuning={17: 0}), False),                               # blend forward: the full kernel instead of the lean one
    "bwd_no_atomics": (dict(_tuning={18: 1}), False),                      # measurement hook: blend backward without its atomics (wrong gradients)
    "pf_off": (dict(_tuning={16: 0}), False),                              # blend kernels: L2 warm-up of the scalar record path off / block of 8, 32, 64 list positions (default 16)
    "pf8": (dict(_tuning={16: 8}), False),
    "pf32": (dict(_tuning={16: 32}), False),
    "pf64": (dict(_tuning={16: 64}), False),
    "proj_early": (dict(_tuning={12: 1}), False),                          # fused projection: SH loads in front of the tile walk (csrc/fused.hip)
    "bwd_sp": (dict(_tuning={5: 2}), False),                               # blend backward: the splat-parallel formulation (csrc/raster.hip raster_backward_sp_kernel)
    "stat_epoch": (dict(), True),
    "stat_epoch_tile": (dict(long_list_global=0), True),
}


DEFAULTS = {}


def load_recipe():
    """path == "recipe": the state bench.py's training_state leg reaches (teacher -> student on the bench scene and its 8 cameras, 60 epochs
    of the reference schedule: 1.6 s) instead of a stored cloud of the 150-camera run -- the same regime (23 M instances per frame)"""
    global RUN_FRAMES
    from bench import build_training_state
    n, W_, H_, f_ = S.CONFIGS["3m_1080p"]
    RUN_FRAMES = 8
    tr, _ = build_training_state(n, W_, H_, f_, S.make_scene(n, seed=0), 8, int(os.environ.get("LATE_RECIPE_EPOCHS", "60")))
    pick = list(range(8))
    order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
    by = {g["name"]: g["params"][0] for g in tr.opt.param_groups}
    moments = [(tr.opt.state[by[nme]]["exp_avg"].clone(), tr.opt.state[by[nme]]["exp_avg_sq"].clone()) for nme in order]

    def restore():
        for i, nme in enumerate(order):
            tr.opt.state[by[nme]]["exp_avg"].copy_(moments[i][0])
            tr.opt.state[by[nme]]["exp_avg_sq"].copy_(moments[i][1])
        tr.fadam.touched = None
    for g in tr.opt.param_groups:
        g["lr"] = 0.0
    tr.sched.step = lambda: None
    DEFAULTS.update(long_list_global=tr.renderer.long_list_global, stat_schedule_always=tr.renderer.stat_schedule_always)
    return tr, pick, restore, dict(epoch=60, degree=tr.degree)


def load(path):
    if path == "recipe":
        return load_recipe()
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
    DEFAULTS.update(long_list_global=tr.renderer.long_list_global, stat_schedule_always=tr.renderer.stat_schedule_always)
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
    base = dict(long_list_global=DEFAULTS["long_list_global"], depth_order=2, stat_schedule_always=DEFAULTS["stat_schedule_always"], replicas_enabled=True,
                refresh_stat_schedule=False)
    base.update(attrs)
    from litegs_amd._lib import check, lib
    tuning = {5: 1, 8: 0, 10: 256, 11: 0, 12: 0, 15: 8, 16: 16, 17: 1, 18: 0, 19: 0, 20: 0, 22: 0, 23: 9, 24: 1024, 26: 2}                          # lg_set_tuning keys a variant may change, at their defaults
    tuning.update(base.pop("_tuning", {}))
    for key, val in tuning.items():
        check(lib().lg_set_tuning(int(key), int(val)), "lg_set_tuning")
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
        try:                                                 # work units of the blend at this cloud (bench.py's device-side counters)
            sys.path.insert(0, ROOT)
            from bench import frame_units
            u = frame_units(tr, pick[0])
            print(f"units of camera {pick[0]}: visible Gaussians {u['n_vis']}  instances {u['I']}  list entries walked {u['I_vis']}  contributing (tile, splat) {u['I_c']}  "
                  f"(pixel, splat) pairs {u['pairs']}", flush=True)
        except Exception as e:                               # measurement aid only
            print("units: not available:", repr(e), flush=True)
        names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(VARIANTS)
        restore(); configure(tr, {}); timed(tr, pick, 1, False)      # discarded: the first timing of a process has come out 1.6 ms per step slow
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
    elif mode == "critical":
        # Is the blend bound by throughput or by its longest tile?  The whole frame's forward / backward against the same kernels on the K
        # heaviest tiles only (the `tiles` argument of the C ABI): if the heaviest tile alone takes most of the frame's time, the launch is
        # a critical path, not a throughput problem.
        import ctypes
        from litegs_amd._lib import check, lib
        L = lib()
        configure(tr, {})
        k = pick[0]
        with torch.no_grad():
            tr.forward_only(k); tr.forward_only(k)
        torch.cuda.synchronize()
        ws1, N = rd.last_ws1
        ws2, tl, _ = rd.last_ws2
        o_ts = L.lg_fused_tile_start_offset(tl, N, H, W, 8, 16)
        o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), tl, N, H, W, 8, 16)
        ts = ws2[o_ts:o_ts + 4 * (rd.ntiles + 2)].view(torch.int32)
        pts = ws2[o_pts:o_pts + 4 * tl].view(torch.int32)
        packed = ws1[L.lg_fused_packed_offset(N):][:4 * N * 16].view(torch.float32)
        dev = ts.device
        img = torch.empty((1, 3, rd.Hp, rd.Wp), device=dev); trans = torch.empty((1, 1, rd.Hp, rd.Wp), device=dev)
        last = torch.empty((1, 1, rd.Hp, rd.Wp), dtype=torch.int16, device=dev)
        work = torch.zeros((rd.ntiles + 1,), dtype=torch.int32, device=dev)
        sm = torch.cuda.current_stream().cuda_stream

        def fwd(tiles):
            K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
            check(L.lg_raster_forward(pts.data_ptr(), ts.data_ptr(), packed.data_ptr(), tp, K, 1, tl, N, H, W, 8, 16, 0, img.data_ptr(), trans.data_ptr(),
                                      last.data_ptr(), None, None, None, work.data_ptr() if tiles is None else None, sm), "fwd")
        d_img = torch.rand((1, 3, rd.Hp, rd.Wp), device=dev) - 0.5
        pg = torch.zeros((N, 16), device=dev)

        def bwd(tiles):
            K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
            check(L.lg_raster_backward(pts.data_ptr(), ts.data_ptr(), packed.data_ptr(), tp, K, trans.data_ptr(), last.data_ptr(), d_img.data_ptr(), None,
                                       1, tl, N, H, W, 8, 16, 0, pg.data_ptr(), None, None, None, sm), "bwd")

        def timeit(fn, arg, reps=5):
            fn(arg); torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(arg); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            return best
        fwd(None); torch.cuda.synchronize()
        walked = work[1:].cpu().numpy()
        order = np.argsort(-walked)
        q = np.percentile(walked, [50, 90, 99, 99.9])
        print(f"splats walked per tile (forward, camera {k}): sum {int(walked.sum())}  mean {walked.mean():.0f}  p50 {q[0]:.0f}  p90 {q[1]:.0f}  p99 {q[2]:.0f}  p99.9 {q[3]:.0f}  max {walked.max()}", flush=True)
        t_all_f, t_all_b = timeit(fwd, None), timeit(bwd, None)
        print(f"whole frame: forward {t_all_f * 1e3:8.1f} us   backward {t_all_b * 1e3:8.1f} us", flush=True)
        for K in (1, 4, 16, 64, 256, 1024, 4096):
            tl_ = torch.from_numpy((order[:K] + 1).astype(np.int32)[None].copy()).to(dev)
            print(f"{K:5d} heaviest tiles ({int(walked[order[:K]].sum()):9d} entries walked, heaviest {int(walked[order[0]])}): forward {timeit(fwd, tl_) * 1e3:8.1f} us   "
                  f"backward {timeit(bwd, tl_) * 1e3:8.1f} us", flush=True)
        fwd(None)
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
        # Two checks per camera and list-building mode, on a cloud 18 000 training iterations away from the synthetic test scenes:
        #  (1) EXACT: the oracle's binning (get_allocate_size -> stable depth order -> create_table -> tile_range) fed with the executor's OWN
        #      per-splat records (read back from workspace 1) must reproduce the executor's tile counts, range table and depth-ordered lists
        #      bit for bit; the oracle's blend of that table against the executor's image within the pinned flip allowance;
        #  (2) END TO END: the oracle's whole pipeline from the raw parameters (its own exp / normalize, CPU math library): instance count
        #      within 2e-6, image within the flip allowance (tile counts are integer functions of floats: they may differ by a few).
        import ctypes
        from litegs_amd._lib import lib
        from oracle import oracle as O
        L = lib()
        scene = [p.detach().cpu().numpy() for p in tr.params]
        bad = 0
        for attrs_name in ("default", "global"):
            configure(tr, VARIANTS[attrs_name][0])
            rd.stat_schedule_always = False                 # the executor's own path end to end (tables are the same either way)
            for k in pick[:int(os.environ.get("LATE_PARITY_FRAMES", "2"))]:
                fr = tr.frames[k]
                with torch.no_grad():
                    img = tr.forward_only(k)
                torch.cuda.synchronize()
                ws1, N = rd.last_ws1
                ws2, tl, _ = rd.last_ws2
                rec = ws1[L.lg_fused_packed_offset(N):][:4 * N * 16].view(torch.float32).view(N, 16).cpu().numpy()
                alloc_x = ws1[L.lg_fused_alloc_offset(N):][:4 * N].view(torch.int32).cpu().numpy()
                emitted = alloc_x > 0
                total = int(rd.fb_total[k])
                o_ts = L.lg_fused_tile_start_offset(tl, N, H, W, 8, 16)
                o_pts = L.lg_fused_sorted_points_offset(ctypes.byref(rd.last_ctx), tl, N, H, W, 8, 16)
                ts = ws2[o_ts:o_ts + 4 * (rd.ntiles + 2)].view(torch.int32).cpu().numpy()
                pts = ws2[o_pts:o_pts + 4 * total].view(torch.int32).cpu().numpy()
                # (1) the oracle's binning on the executor's records (records of splats that emit nothing are not written: mask them out)
                ndc = np.zeros((1, 4, N), np.float32); ndc[0, 0] = rec[:, 13]; ndc[0, 1] = rec[:, 14]
                vz = np.where(emitted, rec[:, 12], np.float32(3.0e38)).astype(np.float32)[None]
                inv = np.zeros((1, 2, 2, N), np.float32); inv[0, 0, 0] = rec[:, 9]; inv[0, 0, 1] = rec[:, 10]; inv[0, 1, 0] = rec[:, 10]; inv[0, 1, 1] = rec[:, 11]
                op = np.where(emitted, rec[:, 5], np.float32(0.0)).astype(np.float32)[None]
                ndc[0, :2, ~emitted] = 0.0; inv[0][:, :, ~emitted] = 0.0
                _, _, alloc_o = O.get_allocate_size(ndc, vz, inv, op, H, W, 8, 16)
                alloc_o = np.where(emitted, alloc_o[0], 0)
                same_alloc = np.array_equal(alloc_o, np.where(emitted, alloc_x, 0))
                dsi = np.argsort(vz, axis=-1, kind="stable").astype(np.int64)
                prefix = np.cumsum(np.take_along_axis(alloc_o[None], dsi, axis=-1), axis=-1, dtype=np.int64).astype(np.int32)
                st_o, spt_o, _, _ = O.create_table(ndc, inv, op, prefix, dsi, H, W, 8, 16)
                ts_o = O.tile_range(st_o, rd.ntiles)
                same_total = int(prefix[0, -1]) == total
                same_ts = same_total and np.array_equal(ts, ts_o[0])
                same_pts = same_total and np.array_equal(pts, spt_o[0][:total])
                rec_o = np.zeros((1, N, 16), np.float32)    # the oracle's record layout (oracle/litegs_oracle.c orc_pack_params): px py a b c r g b opacity depth
                for dst, src in enumerate((0, 1, 9, 10, 11, 6, 7, 8, 5, 12)):
                    rec_o[0, :, dst] = np.where(emitted, rec[:, src], np.float32(0.0))
                img_o, *_ = O.raster_forward(spt_o, ts_o, rec_o, H, W, 8, 16)
                err1 = np.abs(img.cpu().numpy() - np.clip(img_o[..., :H, :W], 0, 1))
                flips1 = int((err1 > 1e-4).sum())
                print(f"parity {attrs_name:8s} camera {k:3d} [same inputs]  instances {total}  tile counts identical {same_alloc}  range table identical {same_ts}  "
                      f"lists identical {same_pts}  image max |d| {err1.max():.2e}  pixels beyond 1e-4: {flips1} of {err1.size}", flush=True)
                bad += 0 if (same_alloc and same_ts and same_pts and flips1 <= 250) else 1
                # (2) end to end from the raw parameters
                ref = O.render_forward(scene, fr.view.cpu().numpy(), fr.proj.cpu().numpy(), fr.planes.cpu().numpy(), H, W, tr.degree)
                ok_n = abs(total - ref.n_instances) <= max(2, int(2e-6 * ref.n_instances))
                err = np.abs(img.cpu().numpy() - np.clip(ref.img[..., :H, :W], 0, 1))
                flips = int((err > 1e-4).sum())
                print(f"parity {attrs_name:8s} camera {k:3d} [end to end]   instances {total} (oracle {ref.n_instances}, within 2e-6: {ok_n})  "
                      f"image max |d| {err.max():.2e}  pixels beyond 1e-4: {flips} of {err.size}", flush=True)
                bad += 0 if (ok_n and flips <= 250) else 1
        print("LATE_PHASE_PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
    tr.close()


if __name__ == "__main__":
    main()
