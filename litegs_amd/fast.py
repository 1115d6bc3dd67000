"""Fused fast path: ``render_preprocess`` + ``render`` (+ their backward) as three native calls.

Same operator semantics as ``litegs_amd.render`` (which mirrors litegs/render/__init__.py:11-94 call by call) but
executed by the native executor of csrc/fused.hip: one fused per-Gaussian kernel instead of eight operators, no
intermediate tensors, all launches of a stage enqueued by ONE C call.  The GPU-driven sizing protocol is the
reference's (litegs/data.py:236-241, GR/compact.cu:527-546, GR/binning.cu:139-163): per-frame pinned feedback buffers
written by an async 4-byte copy in step k and read in step k+1 give the allocation sizes (1.2x visible chunks,
1.5x tile instances); only the first visit of a frame takes a blocking read.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import torch

from ._lib import check, lib
from .statistics import STATS
from .wrapper import CompactedTensor


_GUARD_ALLOC = os.environ.get("LITEGS_GUARD_ALLOC", "0") == "1"


def _empty(shape, dtype, device, zero: bool = False, align: int = 16):
    """torch.empty / torch.zeros for the executor's per-frame buffers.  LITEGS_GUARD_ALLOC=1 (debugging aid; pair it with
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 so that every tensor is a device mapping of its own): the tensor is placed so that it ENDS where its
    page-granular allocation ends (`align`-byte granularity), which turns a read or write past the end of a buffer into an immediate
    memory access fault instead of a silent access to a neighbour."""
    if not _GUARD_ALLOC:
        return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    padded = (nbytes + align - 1) // align * align
    total = max((padded + 4095) // 4096 * 4096, 4096)
    raw = (torch.zeros if zero else torch.empty)((total,), dtype=torch.uint8, device=device)
    return raw[total - padded: total - padded + nbytes].view(dtype).view(*shape)


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


class CameraFrame:
    """Device + host copies of one camera (the kernels take the 4x4 matrices by value)."""

    def __init__(self, view: torch.Tensor, proj: torch.Tensor, planes: torch.Tensor, index: int):
        self.view, self.proj, self.planes, self.index = view, proj, planes, index
        self.view_host = np.ascontiguousarray(view.detach().cpu().numpy().reshape(-1)[:16], dtype=np.float32)
        self.proj_host = np.ascontiguousarray(proj.detach().cpu().numpy().reshape(-1)[:16], dtype=np.float32)
        self.view_ptr = self.view_host.ctypes.data
        self.proj_ptr = self.proj_host.ctypes.data


def _apply_env_options() -> None:
    """LITEGS_DEPTH_ORDER=global|tile|auto: depth sort of all visible splats before the emission (the reference's structure), per-tile
    depth sort after the tile sort (csrc/tilesort.hip: no sort over the splats), or (default) per frame by its size; same tables bit
    for bit (csrc/fused.hip)"""
    mode = os.environ.get("LITEGS_DEPTH_ORDER")
    if mode is not None:
        if mode not in ("tile", "global", "auto"):
            raise ValueError("LITEGS_DEPTH_ORDER must be 'global', 'tile' or 'auto'")
        check(lib().lg_fused_set_option(0, {"global": 0, "tile": 1, "auto": 2}[mode]), "set_option")
    scatter = os.environ.get("LITEGS_TILE_SCATTER")       # 1: group by tile with counts + cursors in the per-tile mode; 0: stable tile radix sort
    if scatter is not None:
        check(lib().lg_fused_set_option(2, 1 if scatter != "0" else 0), "set_option")


class FusedRenderer:
    def __init__(self, n_frames: int, height: int, width: int, tile=(8, 16), cluster_size: int = 128):
        _apply_env_options()
        self.H, self.W, self.TH, self.TW, self.S = height, width, tile[0], tile[1], cluster_size
        self.fb_vis = torch.zeros((n_frames,), dtype=torch.int32).pin_memory()
        self.fb_total = torch.zeros((n_frames,), dtype=torch.int32).pin_memory()
        self.Hp = (height + tile[0] - 1) // tile[0] * tile[0]
        self.Wp = (width + tile[1] - 1) // tile[1] * tile[1]
        self.last_sizes = (0, 0)
        # Per frame: a heaviest-first tile schedule (csrc/raster.hip; a hint -- results do not depend on it -- recomputed on a frame's
        # first visit and then every `cull_refresh`-th) and two depth-bound blocks (csrc/lg_tilewalk.h) used alternately: every
        # visit's blend forward records per-tile saturation depths, and the next visit of the frame skips the splats no tile will
        # reach (depth-bound culling, csrc/fused.hip); a gated fallback keeps the result exact.  Every `cull_refresh`-th visit of a
        # frame runs unculled (fresh bounds from the complete lists, fresh full table size, fresh schedule).
        self.ntiles = (self.Hp // tile[0]) * (self.Wp // tile[1])
        self.n_frames = n_frames
        self.sched = None
        self.sched_cur = [0] * n_frames
        self.sched_valid = [False] * n_frames
        self.tile_order = None
        self.tile_order_valid = [False] * n_frames
        self.cull_enabled = os.environ.get("LITEGS_DEPTH_CULL", "1") != "0"
        self.cull_refresh = 16
        # depth-order mode 'tile' only.  Measured (3 M @1080p): interleaving brings the queue kernel back to its depth-order time
        # (35 -> 18 us) but slows the in-workgroup kernel (66 -> 72 us) and the now gathered scan (16 -> 26 us): no gain, off by default
        self.interleave_emission = os.environ.get("LITEGS_EMISSION_ORDER", "ids") == "interleaved"
        # margin of the depth bounds (csrc/raster.hip: percent of the splats walked beyond a tile's saturation point), per frame.  A
        # fallback costs a whole second binning + blend (~0.4 ms at 3 M @1080p), a wider margin only a few more instances (~35 us per
        # million): measured over 40 training steps of the bench scene, margin 50 % -> 12 fallbacks, 1.093 ms/step; 100 % -> 1 fallback,
        # 0.986 ms; 200 % -> none, 1.026 ms (gpurun_out/margin_ab.log).  Base 100 %; the margin of a frame doubles when its previous
        # visit fell back and decays back after clean visits.  LITEGS_CULL_MARGIN=<percent> pins it.
        pin = os.environ.get("LITEGS_CULL_MARGIN")
        self.margin_fixed = int(pin) if pin else 0
        self.margin_lo, self.margin_hi = 100, 400
        self.margin = [self.margin_fixed or self.margin_lo] * n_frames
        self.clean_visits = [0] * n_frames
        # visits a frame renders unculled after one of its depth bounds was violated (0 = cull again at once, the measured default);
        # LITEGS_CULL_COOLDOWN=<visits>: an unmeasured knob for many-camera runs, where a frame's bounds age 100+ steps between visits
        self.cull_cooldown = int(os.environ.get("LITEGS_CULL_COOLDOWN", "0"))
        self.cooldown = [0] * n_frames
        self.margin_written = [self.margin[0]] * n_frames    # margin of the bounds a frame's next visit will cull with
        self.margin_emitted = [self.margin[0]] * n_frames    # margin of the bounds behind the frame's last emitted total (fb_total)
        self.fallbacks = 0                                   # visits that were re-run unculled (observed one visit later)
        self.last_capacity = [0] * n_frames                  # table capacity a frame's last unculled visit ran with
        self.truncated_visits = 0                            # unculled visits whose table turned out too short (observed one visit later)
        self.visits = [0] * n_frames
        self.full_total = [0] * n_frames                     # host copy of the full table length of a frame's last unculled visit
        self.last_unculled = [False] * n_frames
        self.fb_full = torch.zeros((n_frames,), dtype=torch.int32).pin_memory()     # written by the device when a fallback ran
        self.last_cull = False
        # fuse_optimizer: backward stops after the blend backward; FusedAdam.step() then runs the per-Gaussian backward fused
        # with the Adam update (csrc/fused.hip: project_backward_adam_kernel) -- parameter gradients never go to HBM.
        # Only valid when nothing needs the gradients between backward and the optimizer step (no DP exchange).
        self.fuse_optimizer = False
        self.pending = None
        self.probe_events = None      # measurement hook (bench.py): a list that receives an event pair around every blend backward launch
        # data-parallel hook (dp.MomentExchange.begin): called with (visible_chunkid, visible_chunks_num) as soon as the culling is
        # enqueued, so that the union-of-visibility collective runs on RCCL's stream underneath the whole forward + blend backward
        self.after_cull = None
        self._cull_scratch, self._cull_chunks, self._cull_epoch = None, -1, 0
        # speculative culling (csrc/fused.hip): set by FrameTrainer when it takes over the replay of failed steps
        # gradient replicas (csrc/raster.hip): splats that cover many tiles get several gradient lines; on for renders whose records only
        # the fused backward kernels consume (no statistics, no data-parallel exchange).  LITEGS_GRAD_REPLICAS=0 disables.
        self.replicas_enabled = os.environ.get("LITEGS_GRAD_REPLICAS", "1") != "0"
        self.stat_schedule_always = os.environ.get("LITEGS_STAT_TILE_SCHEDULE", "always") != "stat"
        # Experimental, off by default (unmeasured): in the automatic depth-order mode, a frame whose previous visit emitted more than this
        # many instances per tile on average takes the splat sort + stable tile radix sort (17 us per million instances + 75 us) instead of
        # tile scatter + per-tile sort, whose long-list regimes cost 30-34 us per million (profiles/r03_tilesort_scaling.log).
        self.long_list_global = int(os.environ.get("LITEGS_LONG_LIST_GLOBAL", "0"))
        self.hot_counter = None
        self.spec = None              # dict(poison=device int32[1], poison_host / applied_host = pinned int32[1])
        self.spec_step = 0            # number of the training step being enqueued
        self.force_full = False       # the next render runs unculled (the first replayed step)

    def enable_speculation(self, device):
        if self.spec is None:
            self.spec = dict(poison=torch.zeros((1,), dtype=torch.int32, device=device),
                             poison_host=torch.zeros((1,), dtype=torch.int32).pin_memory(),
                             applied_host=torch.zeros((1,), dtype=torch.int32).pin_memory())
        return self.spec

    def speculation_args(self, active: bool):
        """arguments of lg_fused_set_speculation for the coming call (all NULL: the gated repeat / an unconditional Adam)"""
        if active and self.spec is not None:
            sp = self.spec
            return (sp["poison"].data_ptr(), sp["poison_host"].data_ptr(), sp["applied_host"].data_ptr(), int(self.spec_step))
        return (None, None, None, 0)

    def reset_feedback(self):
        """parameters were replaced / re-sorted: forget everything predicted from earlier visits (sizes, schedules, depth bounds)"""
        self.fb_vis.zero_(); self.fb_total.zero_(); self.fb_full.zero_()
        n = self.n_frames
        self.sched_valid = [False] * n
        self.tile_order_valid = [False] * n
        self.full_total = [0] * n
        self.last_unculled = [False] * n
        self.last_capacity = [0] * n
        self.margin = [self.margin_fixed or self.margin_lo] * n
        self.clean_visits = [0] * n
        self.cooldown = [0] * n
        self.margin_written = [self.margin[0]] * n
        self.margin_emitted = [self.margin[0]] * n

    def emission_order(self, A: int, S: int, device):
        """depth-order mode 'tile' only: slot j -> splat ((j mod A) * P mod A) * S + j div A with P coprime to A -- every group of 256
        consecutive slots draws one splat from each of 256 chunks that lie far apart in the (Morton-ordered) cloud.  Cached per A."""
        import math
        cache = self.__dict__.setdefault("_emit_cache", {})
        key = (A, S)
        if key not in cache:
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            P = max(int(A * 0.6180339887) | 1, 1)
            while math.gcd(P, A) != 1:
                P += 2
            j = np.arange(A * S, dtype=np.int64)
            order = ((j % A) * P % A) * S + j // A
            cache[key] = torch.from_numpy(order.astype(np.int32)).to(device)
        return cache[key]

    def cull_scratch(self, chunks: int, device):
        """persistent look-back table of the multi-workgroup culling kernel (epoch-tagged: zeroed once, never cleared again)"""
        if self._cull_scratch is None or self._cull_chunks != chunks or self._cull_epoch > 1_000_000:
            self._cull_scratch = torch.zeros((lib().lg_fused_cull_scratch_bytes(chunks),), dtype=torch.uint8, device=device)
            self._cull_chunks, self._cull_epoch = chunks, 0
        self._cull_epoch += 1
        return self._cull_scratch.data_ptr(), self._cull_epoch

    def render(self, frame: CameraFrame, cluster_origin, cluster_extend, xyz, scale, rot, sh_0, sh_rest, opacity, degree: int):
        """-> (img[1,3,H,W] clamped to [0,1], visible_chunkid, visible_chunks_num)."""
        img, vis_ids, vis_num = _RenderFn.apply(self, frame, cluster_origin, cluster_extend, degree, xyz, scale, rot, sh_0, sh_rest, opacity)
        img = img[..., : self.H, : self.W].clamp(0, 1)
        return img, vis_ids, vis_num

    def render_raw(self, frame: CameraFrame, cluster_origin, cluster_extend, xyz, scale, rot, sh_0, sh_rest, opacity, degree: int):
        """-> (raw tile-padded image [1,3,Hp,Wp] as the blend kernel wrote it, visible_chunkid, visible_chunks_num); pair it with
        loss_hip.raster_l1_ssim_loss, which applies the crop and the clamp inside the loss kernels."""
        return _RenderFn.apply(self, frame, cluster_origin, cluster_extend, degree, xyz, scale, rot, sh_0, sh_rest, opacity)


def _d_opacity(pg, ws1, N):
    """gradient w.r.t. the ACTIVATED opacity from the blend backward's moment record: slot 8 holds sum(m) = opacity * d_opacity
    (csrc/raster.hip); the opacity is dword 5 of the packed splat record at the head of workspace 1's `packed` area."""
    L = lib()
    off = L.lg_fused_packed_offset(N)
    rec = ws1[off:off + 4 * N * L.lg_packed_record_floats()].view(torch.float32).view(N, L.lg_packed_record_floats())
    o = rec[:, 5]
    return torch.where(o > 0, pg[:, 8] / o, torch.zeros_like(o)).reshape(1, 1, N)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R: FusedRenderer, frame: CameraFrame, origin, extend, degree, xyz, scale, rot, sh_0, sh_rest, opacity):
        L = lib()
        dev = xyz.device
        chunks, S = xyz.shape[-2], xyz.shape[-1]
        k = frame.index
        s = _s()
        visibility = _empty((chunks,), torch.bool, dev)
        vis_num = _empty((1,), torch.int32, dev)
        vis_ids = _empty((chunks,), torch.int64, dev)
        fb_vis_ptr = R.fb_vis.data_ptr() + 4 * k
        fb_tot_ptr = R.fb_total.data_ptr() + 4 * k
        common = (origin.data_ptr(), extend.data_ptr(), frame.planes.data_ptr(), chunks, frame.view_ptr, frame.proj_ptr, R.H, R.W, R.TH, R.TW,
                  int(degree), xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), sh_0.data_ptr(), sh_rest.data_ptr(), opacity.data_ptr(), S)
        pred_vis = int(R.fb_vis[k])
        do_cull = 1
        if pred_vis <= 0:                                    # first visit: blocking count (GR/compact.cu:543-546)
            check(L.lg_fused_stage1(*common, 1, visibility.data_ptr(), vis_num.data_ptr(), vis_ids.data_ptr(), 0, None, 0, fb_vis_ptr, None,
                                    *R.cull_scratch(chunks, dev), None, None, s),
                  "fused cull")
            A = int(vis_num.item())
            do_cull = 0
        else:
            A = min(int(1.2 * pred_vis), chunks)
        A = max(A, 1)
        N = A * S
        stat = STATS.active
        replicas = bool(R.replicas_enabled and R.fuse_optimizer and R.after_cull is None and not stat and any(ctx.needs_input_grad))
        if replicas and R.hot_counter is None:
            R.hot_counter = _empty((1,), torch.int32, dev, zero=True)
        if R.hot_counter is not None:
            L.lg_fused_set_hot_counter(R.hot_counter.data_ptr())
        L.lg_fused_set_option(3, 1 if replicas else 0)
        if R.interleave_emission and L.lg_fused_get_option(0) != 0:
            L.lg_fused_set_emission_order(R.emission_order(A, S, dev).data_ptr(), N)
        ws1_bytes = L.lg_fused_workspace1_bytes(N)
        ws1 = _empty((ws1_bytes,), torch.uint8, dev, align=64)            # 64-byte records read by 64-byte scalar loads
        # The reference rasterises along the statistics helper's cached heavy-first tile list whenever a frame has one, i.e. in every render
        # after the first statistics epoch (litegs/render/__init__.py:75-79), and so does the executor by default: from then on its own
        # schedule, depth bounds and speculative culling are idle.  The list is a permutation of ALL tiles -- a schedule, not a selection --
        # so LITEGS_STAT_TILE_SCHEDULE=stat keeps the executor's own machinery outside statistics renders (same image,
        # test_gpu_stats.py).  Measured over the first 4500 iterations of tests/convergence_3m.py (150 cameras: a frame is revisited
        # after 150 steps of a fast-changing cloud, bounds are violated often): 1.66 ms / iteration against 1.49 with the reference's
        # behaviour -- hence not the default.
        tiles = STATS.schedule_for_current_frame() if (stat or R.stat_schedule_always) else None
        # depth-bound culling: bookkeeping of the sizing feedback (the emitted total of a culled visit is not the full table length)
        pred_total = int(R.fb_total[k])
        if R.last_unculled[k] and pred_total > R.last_capacity[k] > 0:
            # the previous (unculled) visit needed more entries than its predicted table held: its tail was dropped, as in the reference
            # (GR/binning.cu:63, silent there).  Counted here, and this visit sizes its table exactly (the blocking first-visit path).
            R.truncated_visits += 1
            pred_total = 0
            R.fb_total[k] = 0
        if R.last_unculled[k] and pred_total > 0:
            R.full_total[k] = pred_total                      # the previous visit of this frame emitted everything
        if int(R.fb_full[k]) > 0:                             # ... or a fallback re-ran it in full
            if int(R.fb_full[k]) > R.last_capacity[k] > 0:    # ... into a table that was too short for it: same treatment
                R.truncated_visits += 1
                pred_total = 0
                R.fb_total[k] = 0
            R.full_total[k] = max(R.full_total[k], int(R.fb_full[k]))
            R.fb_full[k] = 0
            R.fallbacks += 1
            R.clean_visits[k] = 0
            R.cooldown[k] = R.cull_cooldown
            if not R.margin_fixed:
                R.margin[k] = min(R.margin[k] * 2, R.margin_hi)
        elif not R.margin_fixed:
            R.clean_visits[k] += 1
            if R.clean_visits[k] >= 6 and R.margin[k] > R.margin_lo:
                R.margin[k] = max(R.margin_lo, (R.margin[k] * 3) // 4)
                R.clean_visits[k] = 0
        use_sched = not stat and tiles is None
        if use_sched and R.sched is None:
            R.sched = _empty((R.n_frames, 2, L.lg_sched_words(R.H, R.W, R.TH, R.TW)), torch.int32, dev)
            R.tile_order = _empty((R.n_frames, R.ntiles), torch.int32, dev)
        in_ptr = out_ptr = None
        if use_sched:
            cur = R.sched_cur[k]
            if R.sched_valid[k]:
                in_ptr = R.sched[k, cur].data_ptr()
            out_ptr = R.sched[k, 1 - cur].data_ptr()
        refresh = R.visits[k] % R.cull_refresh == 0
        cull = bool(R.cull_enabled and in_ptr is not None and R.full_total[k] > 0 and pred_total > 0 and not refresh and not R.force_full
                    and R.cooldown[k] == 0)
        if R.cooldown[k] > 0:
            R.cooldown[k] -= 1
        R.force_full = False
        order_ptr = (R.tile_order.data_ptr() + 4 * R.ntiles * k) if use_sched else None
        order_in = order_ptr if (use_sched and R.tile_order_valid[k]) else None
        order_out = order_ptr if (use_sched and (refresh or not R.tile_order_valid[k])) else None
        R.visits[k] += 1
        # per-frame override of the depth-order mode (see long_list_global): the same value must hold for stage 1, stage 2 and the backward
        mode_override = None
        if R.long_list_global > 0 and pred_total > R.long_list_global * R.ntiles and L.lg_fused_get_option(0) == 2:
            mode_override = 0
            L.lg_fused_set_option(0, 0)
        check(L.lg_fused_stage1(*common, do_cull, visibility.data_ptr(), vis_num.data_ptr(), vis_ids.data_ptr(), A, ws1.data_ptr(), ws1_bytes,
                                fb_vis_ptr if do_cull else None, fb_tot_ptr, *(R.cull_scratch(chunks, dev) if do_cull else (None, 0)),
                                in_ptr if cull else None, out_ptr, s), "fused stage1")
        if R.after_cull is not None and any(ctx.needs_input_grad):
            R.after_cull(vis_ids, vis_num)
        if pred_total <= 0:                                  # first visit: blocking table size (GR/binning.cu:152-163)
            off = L.lg_fused_total_offset(N)
            table_len = int(ws1[off:off + 4].view(torch.int32).item())
        elif cull:
            table_len = int(1.5 * R.full_total[k])           # capacity for the fallback's full table
        else:
            table_len = int(1.5 * max(pred_total, R.full_total[k]))
        table_len = max(table_len, 1)
        # the emitted total was predicted under the margin of the visit before last: scale the culled table when the bounds got wider
        grow = max(1.0, R.margin_written[k] / max(R.margin_emitted[k], 1))
        len_cull = min(table_len, int(1.5 * grow * pred_total) + 65536) if cull else table_len
        ws2_bytes = L.lg_fused_workspace2_bytes(table_len, N, R.H, R.W, R.TH, R.TW)
        ws2 = _empty((ws2_bytes,), torch.uint8, dev)
        img = _empty((1, 3, R.Hp, R.Wp), torch.float32, dev)
        trans = _empty((1, 1, R.Hp, R.Wp), torch.float32, dev)
        last = _empty((1, 1, R.Hp, R.Wp), torch.int16, dev)
        K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
        fc = fw = None
        if stat:
            fc = _empty((1, 1, N), torch.int32, dev, zero=True)
            fw = _empty((1, 1, N), torch.float32, dev, zero=True)
            STATS.set_compaction(vis_ids[:A], vis_num)
            a_off = L.lg_fused_alloc_offset(N)                # b_visible = allocate_size != 0 (wrapper.py:733-736)
            STATS.add_visible((ws1[a_off:a_off + 4 * N].view(torch.int32) != 0).view(1, N))
        if tiles is not None and tiles.shape[1] != R.ntiles:
            # a partial tile list leaves the other tiles' pixels at "nothing blended".  The statistics helper's cached list is a permutation
            # of all tiles (statistics.py update_tile_schedule: the argsort of the per-tile blend counts): every pixel is written, no fills
            img.zero_(); trans.fill_(1.0); last.zero_()
        # gradient accumulator of the blend backward: allocated here so that stage 2 can clear it on the side (no memset launch later)
        pg_lines = L.lg_fused_grad_lines(N) if replicas else N
        pg = _empty((pg_lines, L.lg_packed_grad_floats()), torch.float32, dev) if any(ctx.needs_input_grad) else None
        L.lg_fused_set_option(1, int(R.margin[k]))
        # a culled render that a fused Adam step follows may run speculatively (no gated repeat); anything else keeps the repeat
        L.lg_fused_set_speculation(*R.speculation_args(cull and R.fuse_optimizer and any(ctx.needs_input_grad)))
        check(L.lg_fused_stage2(A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes, tp, K,
                                1 if stat else 0, img.data_ptr(), trans.data_ptr(), last.data_ptr(),
                                fc.data_ptr() if stat else None, fw.data_ptr() if stat else None,
                                pg.data_ptr() if pg is not None else None,
                                order_in, order_out, in_ptr, out_ptr, 1 if cull else 0, len_cull, R.fb_full.data_ptr() + 4 * k,
                                frame.view_ptr, frame.proj_ptr, int(degree), chunks,
                                xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), sh_0.data_ptr(), sh_rest.data_ptr(), opacity.data_ptr(),
                                vis_ids.data_ptr(), vis_num.data_ptr(), s), "fused stage2")
        if mode_override is not None:
            L.lg_fused_set_option(0, 2)
        ctx.mode_override = mode_override
        if use_sched:
            R.sched_cur[k] = 1 - R.sched_cur[k]
            R.sched_valid[k] = True
        R.last_unculled[k] = not cull
        R.last_capacity[k] = table_len
        R.last_cull = cull
        if cull:
            R.margin_emitted[k] = R.margin_written[k]
        R.margin_written[k] = R.margin[k]
        R.last_ws1 = (ws1, N)                              # for tests: the fallback flag lives in workspace 1 (lg_fused_flags_offset)
        R.last_ws2 = (ws2, table_len, N)                   # for tools: the tile range table lives in workspace 2 (lg_fused_tile_start_offset)
        if order_out is not None:
            R.tile_order_valid[k] = True
        ctx.order_ptr = order_ptr if (use_sched and R.tile_order_valid[k]) else None
        ctx.pg = pg
        ctx.replicas = replicas
        if stat:
            STATS.update_tile_schedule(last, R.TH, R.TW)
        ctx.R, ctx.frame, ctx.meta = R, frame, (A, S, table_len, int(degree), chunks, sh_rest.shape[0], ws1_bytes, ws2_bytes, stat)
        ctx.tiles = tiles
        ctx.stat_bufs = (fc, fw)
        ctx.save_for_backward(ws1, ws2, vis_ids, vis_num, trans, last, xyz, scale, rot, sh_0, sh_rest, opacity)
        ctx.mark_non_differentiable(vis_ids, vis_num)
        ctx.set_materialize_grads(False)          # no zero-filled gradients for the two index outputs
        R.last_sizes = (A, table_len)
        return img, vis_ids[:A], vis_num

    @staticmethod
    def backward(ctx, g_img, _g_ids, _g_num):
        ws1, ws2, vis_ids, vis_num, trans, last, xyz, scale, rot, sh_0, sh_rest, opacity = ctx.saved_tensors
        R, frame = ctx.R, ctx.frame
        A, S, table_len, degree, chunks, Rr, ws1_bytes, ws2_bytes, stat = ctx.meta
        L = lib()
        dev = xyz.device
        N = A * S
        if g_img is None:
            return (None,) * 11
        g_img = g_img.contiguous()
        pg, pg_zero = ctx.pg, 1
        if pg is None:
            pg, pg_zero = _empty((L.lg_fused_grad_lines(N) if ctx.replicas else N, L.lg_packed_grad_floats()), torch.float32, dev), 0
        ctx.pg = None
        L.lg_fused_set_option(3, 1 if ctx.replicas else 0)          # as it was for this frame's stage 1
        if ctx.mode_override is not None:
            L.lg_fused_set_option(0, ctx.mode_override)              # ... and so the depth-order mode (where the blend finds its lists)
        esq = _empty((1, 1, N), torch.float32, dev, zero=True) if stat else None
        tiles = ctx.tiles
        K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
        if R.fuse_optimizer:
            if R.probe_events is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            check(L.lg_fused_backward(A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes,
                                      frame.view_ptr, frame.proj_ptr, degree, chunks, Rr, vis_ids.data_ptr(), vis_num.data_ptr(),
                                      xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), opacity.data_ptr(), tp, K,
                                      trans.data_ptr(), last.data_ptr(), g_img.data_ptr(), None, None, 1 if stat else 0,
                                      pg.data_ptr(), pg_zero, esq.data_ptr() if stat else None, None, None, None, None, None, None, ctx.order_ptr, _s()),
                  "fused blend backward")
            if ctx.mode_override is not None:
                L.lg_fused_set_option(0, 2)
            if R.probe_events is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                R.probe_events.append((ev0, ev1))
            if stat:
                fc, fw = ctx.stat_bufs
                STATS.add_moments("fragment_weight", fw, fw * fw, fc)
                STATS.add_moments("fragment_err", _d_opacity(pg, ws1, N), esq, fc)
            # ws1 rides along: its tile counts tell the fused backward + Adam which gradient records can only be zero
            R.pending = dict(pg=pg, A=A, S=S, frame=frame, degree=degree, chunks=chunks, Rr=Rr, vis_ids=vis_ids, vis_num=vis_num, ws1=ws1,
                             replicas=ctx.replicas)
            return (None,) * 11
        d_pos = _empty((3, A, S), torch.float32, dev)
        d_scale = _empty((3, A, S), torch.float32, dev)
        d_rot = _empty((4, A, S), torch.float32, dev)
        d_sh0 = _empty((3, A, S), torch.float32, dev)
        d_shr = _empty((Rr * 3, A, S), torch.float32, dev)
        d_opa = _empty((1, A, S), torch.float32, dev)
        check(L.lg_fused_backward(A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes,
                                  frame.view_ptr, frame.proj_ptr, degree, chunks, Rr, vis_ids.data_ptr(), vis_num.data_ptr(),
                                  xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), opacity.data_ptr(), tp, K,
                                  trans.data_ptr(), last.data_ptr(), g_img.data_ptr(), None, None, 1 if stat else 0,
                                  pg.data_ptr(), pg_zero, esq.data_ptr() if stat else None,
                                  d_pos.data_ptr(), d_scale.data_ptr(), d_rot.data_ptr(), d_sh0.data_ptr(), d_shr.data_ptr(), d_opa.data_ptr(),
                                  ctx.order_ptr, _s()),
              "fused backward")
        if ctx.mode_override is not None:
            L.lg_fused_set_option(0, 2)
        if stat:
            fc, fw = ctx.stat_bufs
            # d_opacity of the activated opacity = packed_grad slot 8 (rasterize_backward's 4th output)
            d_op_act = _d_opacity(pg, ws1, N)
            STATS.add_moments("fragment_weight", fw, fw * fw, fc)
            STATS.add_moments("fragment_err", d_op_act, esq, fc)
        ids = vis_ids[:A]
        grads = (CompactedTensor(xyz.shape, ids, d_pos), CompactedTensor(scale.shape, ids, d_scale), CompactedTensor(rot.shape, ids, d_rot),
                 CompactedTensor(sh_0.shape, ids, d_sh0), CompactedTensor(sh_rest.shape, ids, d_shr), CompactedTensor(opacity.shape, ids, d_opa))
        return (None, None, None, None, None, *grads)


class FusedAdam:
    """All parameter groups in one launch (csrc/fused.hip: adam_multi_kernel); same update rule as adamUpdate."""

    def __init__(self, optimizer, renderer: Optional["FusedRenderer"] = None):
        self.opt = optimizer
        self.renderer = renderer
        self.groups = optimizer.param_groups
        G = len(self.groups)
        self._arr = ctypes.c_void_p * G
        self._iarr = ctypes.c_int * G
        self._farr = ctypes.c_float * G
        self._ready = False
        self.touched = None
        self.skip_untouched = os.environ.get("LITEGS_ADAM_SKIP_UNTOUCHED", "1") != "0"

    def _init_state(self):
        for g in self.groups:
            p = g["params"][0]
            st = self.opt.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        ps = [g["params"][0] for g in self.groups]
        self.chunks, self.S = ps[0].shape[-2], ps[0].shape[-1]
        self._p = self._arr(*[p.data_ptr() for p in ps])
        self._m = self._arr(*[self.opt.state[p]["exp_avg"].data_ptr() for p in ps])
        self._v = self._arr(*[self.opt.state[p]["exp_avg_sq"].data_ptr() for p in ps])
        self._rows = self._iarr(*[int(p.numel() // (self.chunks * self.S)) for p in ps])
        self._by_name = {g.get("name"): g for g in self.groups}
        self.touched = None                  # rebuilt from the moments before the next fused backward + Adam
        self._ready = True

    def _touched_flags(self) -> torch.Tensor:
        """uint8 [chunks*S]: 0 where BOTH Adam moments of EVERY row of a Gaussian are zero (it never received a gradient): the fused
        backward + Adam skips such Gaussians while their gradient is zero too -- exactly a no-op (csrc/fused.hip).  Maintained by that
        kernel; every other writer of the moments (the other optimizer paths, density control, re-sort, checkpoints) drops the array
        and it is rebuilt from the moments here (one pass over the optimizer state, only on such transitions)."""
        if self.touched is None or self.touched.numel() != self.chunks * self.S:
            flags = None
            for g in self.groups:
                p = g["params"][0]
                st = self.opt.state[p]
                for key in ("exp_avg", "exp_avg_sq"):
                    nz = (st[key].reshape(-1, self.chunks * self.S) != 0).any(dim=0)
                    flags = nz if flags is None else (flags | nz)
            self.touched = flags.to(torch.uint8).contiguous()
        return self.touched

    def _step_fused_backward(self, pend):
        """per-Gaussian backward + Adam in one kernel, from the packed gradients left by the blend backward."""
        order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
        ps = [self._by_name[n]["params"][0] for n in order]
        ms = [self.opt.state[p]["exp_avg"] for p in ps]
        vs = [self.opt.state[p]["exp_avg_sq"] for p in ps]
        lr6 = (ctypes.c_float * 6)(*[float(self._by_name[n]["lr"]) for n in ["xyz", "sh_0", "sh_rest", "opacity", "scale", "rot"]])
        R, fr = self.renderer, pend["frame"]
        lib().lg_fused_set_speculation(*R.speculation_args(True))
        hot = None
        if pend.get("replicas") and "ws1" in pend:
            hot = pend["ws1"].data_ptr() + lib().lg_fused_hot_offset(pend["A"] * pend["S"])
        lib().lg_fused_set_hot_table(hot)
        check(lib().lg_fused_backward_adam(pend["A"], pend["S"], R.H, R.W, fr.view_ptr, fr.proj_ptr, pend["degree"], pend["chunks"], pend["Rr"],
                                           pend["vis_ids"].data_ptr(), pend["vis_num"].data_ptr(), pend["pg"].data_ptr(), None,
                                           *[p.data_ptr() for p in ps], *[m.data_ptr() for m in ms], *[v.data_ptr() for v in vs],
                                           lr6, 0.9, 0.999, float(self.groups[0]["eps"]),
                                           self._touched_flags().data_ptr() if self.skip_untouched else None,
                                           (pend["ws1"].data_ptr() + lib().lg_fused_alloc_offset(pend["A"] * pend["S"])) if "ws1" in pend else None,
                                           _s()), "fused backward+adam")

    @torch.no_grad()
    def step_exchange(self, exchange, cams, slot: int = 0):
        """data-parallel step (litegs_amd/dp.py: MomentExchange): the pending blend-backward moments of this rank are exchanged and the
        per-Gaussian backward + Adam runs over the union of the ranks' visible chunks -> (union_ids, union_count)"""
        if not self._ready:
            self._init_state()
        pend, self.renderer.pending = self.renderer.pending, None
        if pend is None:
            raise RuntimeError("step_exchange: no pending blend backward (FusedRenderer.fuse_optimizer must be on)")
        order = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]
        ps = [self._by_name[n]["params"][0] for n in order]
        ms = [self.opt.state[p]["exp_avg"] for p in ps]
        vs = [self.opt.state[p]["exp_avg_sq"] for p in ps]
        lr6 = [float(self._by_name[n]["lr"]) for n in ["xyz", "sh_0", "sh_rest", "opacity", "scale", "rot"]]
        R = self.renderer
        return exchange.step(pend, cams, ps, ms, vs, lr6, float(self.groups[0]["eps"]), R.H, R.W, slot,
                             self._touched_flags() if self.skip_untouched else None)

    @torch.no_grad()
    def step(self, visible_chunk: torch.Tensor, visible_chunks_num: Optional[torch.Tensor]):
        if not self._ready:
            self._init_state()
        if self.renderer is not None and self.renderer.pending is not None:
            pend, self.renderer.pending = self.renderer.pending, None
            return self._step_fused_backward(pend)
        self.touched = None                  # the paths below update the moments without maintaining the flags
        ps = [g["params"][0] for g in self.groups]
        grads = [p.grad for p in ps]
        if any(g is None for g in grads):
            return self.opt.step(visible_chunk, visible_chunks_num, None)
        dense = not isinstance(grads[0], CompactedTensor)
        if dense:
            gp = self._arr(*[g.data_ptr() for g in grads])
        else:
            gp = self._arr(*[g.compacted_values.data_ptr() for g in grads])
        lr = self._farr(*[float(g["lr"]) for g in self.groups])
        eps = float(self.groups[0]["eps"])
        A = visible_chunk.shape[0]
        check(lib().lg_adam_update_multi(len(ps), self._p, gp, self._m, self._v, self._rows, lr, visible_chunk.data_ptr(),
                                         visible_chunks_num.data_ptr() if visible_chunks_num is not None else None,
                                         self.chunks, A, self.S, 1 if dense else 0, 0.9, 0.999, eps, _s()), "adam_update_multi")
