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


from .hostwords import HostWords

# Environment switches of the executor (each one named in INTEGRATION.md with the measurement behind its default).  Everything else
# that used to be a switch is a plain attribute of FusedRenderer for tests / tools to set.
_GUARD_ALLOC = os.environ.get("LITEGS_GUARD_ALLOC", "0") == "1"
_POISON_ALLOC = os.environ.get("LITEGS_GUARD_ALLOC", "0") == "poison"     # module attribute: tools may set it after import
_DEPTH_ORDER = {"global": 0, "tile": 1, "auto": 2}


def _apply_env_tuning():
    """LITEGS_TUNING="key=value[,key=value...]" (measurement aid): lg_set_tuning launch variants for a whole process (bench.py A/B runs)"""
    spec = os.environ.get("LITEGS_TUNING", "")
    for kv in filter(None, (t.strip() for t in spec.split(","))):
        try:
            key, val = (int(t) for t in kv.split("="))
        except ValueError:
            raise ValueError(f"LITEGS_TUNING: malformed entry {kv!r} (expected key=value with integer key and value)") from None
        check(lib().lg_set_tuning(key, val), f"lg_set_tuning({kv})")


_TUNING_APPLIED = False


class LgFusedCtx(ctypes.Structure):
    """include/litegs_hip.h: the executor's per-call context (the library keeps no process-wide executor state)"""
    _fields_ = [("struct_bytes", ctypes.c_int32), ("depth_order", ctypes.c_int32), ("bound_margin_pct", ctypes.c_int32),
                ("tile_scatter", ctypes.c_int32), ("grad_replicas", ctypes.c_int32), ("step_id", ctypes.c_int32),
                ("debug_validate", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("hot_counter", ctypes.c_void_p), ("poison", ctypes.c_void_p), ("poison_host", ctypes.c_void_p),
                ("applied_host", ctypes.c_void_p), ("debug_words", ctypes.c_void_p)]


def _empty(shape, dtype, device, zero: bool = False, align: int = 16):
    """torch.empty / torch.zeros for the executor's per-frame buffers.  LITEGS_GUARD_ALLOC=1 (debugging aid; pair it with
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 so that every tensor is a device mapping of its own): the tensor is placed so that it ENDS where its
    page-granular allocation ends (`align`-byte granularity), which turns a read or write past the end of a buffer into an immediate
    memory access fault instead of a silent access to a neighbour."""
    if _POISON_ALLOC and not zero:
        # LITEGS_GUARD_ALLOC=poison (debugging aid, works with the caching allocator): every per-frame buffer starts as 0xC1 bytes -- as int32
        # a large negative number, as float -24.2 -- so that a word no kernel of this frame wrote is out of range wherever it is used as a
        # key, an id or a position, in a process's FIRST trainer as well; pair it with LITEGS_VALIDATE_TABLES=1 (profiles/r04_fault_attribution.md)
        t = torch.empty(shape, dtype=dtype, device=device)
        t.view(-1).view(torch.uint8).fill_(0xC1)
        return t
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


class _FrameState:
    """what the executor remembers about one camera between two visits (sizing feedback lives in the pinned words)"""
    __slots__ = ("sched_cur", "sched_valid", "order_valid", "margin", "clean_visits", "margin_written", "margin_emitted",
                 "last_capacity", "visits", "full_total", "last_unculled", "visits_at_replace", "idle_replacements")

    def __init__(self, margin: int):
        self.sched_cur = 0
        self.visits = 0
        self.visits_at_replace = 0
        self.idle_replacements = 0
        self.reset(margin)

    def reset(self, margin: int):
        self.sched_valid = False          # the frame's depth-bound block of its previous visit is usable
        self.order_valid = False          # ... and its heaviest-first tile schedule
        self.margin = margin              # margin (percent) of the bounds this visit writes
        self.clean_visits = 0
        self.margin_written = margin      # margin of the bounds the frame's next visit will cull with
        self.margin_emitted = margin      # margin of the bounds behind the frame's last emitted total (fb_total)
        self.last_capacity = 0            # table capacity of the frame's last visit
        self.full_total = 0               # full table length of the frame's last unculled visit
        self.last_unculled = False


class FusedRenderer:
    """One per trainer / evaluator.  Owns every piece of state the C executor needs between calls -- the library itself is stateless
    (LgFusedCtx) -- and the pinned words the device stores into (library arena: never unmapped, see hostwords.py)."""

    def __init__(self, n_frames: int, height: int, width: int, tile=(8, 16), cluster_size: int = 128):
        global _TUNING_APPLIED
        if not _TUNING_APPLIED:
            _TUNING_APPLIED = True
            _apply_env_tuning()
        self.H, self.W, self.TH, self.TW, self.S = height, width, tile[0], tile[1], cluster_size
        self.Hp = (height + tile[0] - 1) // tile[0] * tile[0]
        self.Wp = (width + tile[1] - 1) // tile[1] * tile[1]
        self.ntiles = (self.Hp // tile[0]) * (self.Wp // tile[1])
        self.n_frames = n_frames
        # GPU-driven sizing feedback, three pinned words per frame, written by device stores in visit k and read in visit k + 1:
        # visible chunks | emitted table length | full table length when a gated fallback ran
        self._words = HostWords(3 * n_frames)
        self.fb_vis = self._words.a[0:n_frames]
        self.fb_total = self._words.a[n_frames:2 * n_frames]
        self.fb_full = self._words.a[2 * n_frames:3 * n_frames]
        self.last_sizes = (0, 0)
        # ---- options (attributes; the environment only selects the first three)
        mode = os.environ.get("LITEGS_DEPTH_ORDER", "auto")
        if mode not in _DEPTH_ORDER:
            raise ValueError("LITEGS_DEPTH_ORDER must be 'global', 'tile' or 'auto'")
        self.depth_order = _DEPTH_ORDER[mode]          # csrc/fused.hip: how each tile's list gets its depth order
        self.cull_enabled = os.environ.get("LITEGS_DEPTH_CULL", "1") != "0"
        # After the first statistics epoch the reference rasterises along the statistics helper's cached tile list (render/__init__.py:75-79).
        # True follows it (the executor's own schedule, depth bounds and speculative culling are then idle); False keeps the executor's
        # machinery outside statistics renders -- same image (test_gpu_stats.py).  Whole-run A/B at 3 M / 150 cameras / 30 000 iterations
        # (profiles/r04_convergence_3m_own_schedule.md vs r04_convergence_3m.md): 3.63-3.65 ms per iteration against 3.57-3.66 -- 1230-1300
        # violated bounds per run (a frame is revisited after 150 steps of a moving cloud) eat what the culling saves, except in the last
        # 40 epochs (3.32 vs 3.49).  Depth-bound culling is a few-camera / converged-cloud feature: the reference's behaviour stays the default.
        self.stat_schedule_always = True
        # The helper's list is re-made in statistics epochs only (statistic_helper.py:68-79): every 5th epoch.  True re-orders it after
        # EVERY render from that render's own last_contributor (two small launches): the order is a hint, the set of tiles the same.
        self.refresh_stat_schedule = False
        self.validate_tables = os.environ.get("LITEGS_VALIDATE_TABLES", "0") == "1"      # debugging aid (csrc/fused.hip "Table validators")
        self.tile_scatter = True           # per-tile mode: group by tile with counts + cursors (False: stable tile radix sort); same tables
        self.replicas_enabled = True       # gradient replicas (csrc/raster.hip) for renders whose records only the fused backward kernels consume
        # Automatic depth-order mode: a frame whose previous visit emitted more than this many instances per tile on average takes the splat
        # sort + stable tile radix sort instead of tile scatter + per-tile sort (0 = never).  The per-tile sort is a one-wave register radix
        # sort up to 1024 entries (8-12 us per million instances) and falls back to bitonic networks beyond (30-34 us per million); late in
        # a density-control run lists average 1200-2500 entries (19-24 M instances per 1080p frame) and the per-tile sort alone is 0.76 ms of
        # a 4.0 ms step, the global route 0.49 ms cheaper in total (profiles/r04_late_phase_ab_tile_vs_global_vs_regime_w.log).  At the
        # bench's clouds (170-720 per tile) the per-tile route wins.
        self.long_list_global = 640
        # Per frame: a heaviest-first tile schedule (csrc/raster.hip; a hint -- results do not depend on it -- recomputed on a frame's
        # first visit and then every `cull_refresh`-th) and two depth-bound blocks (csrc/lg_tilewalk.h) used alternately: every
        # visit's blend forward records per-tile saturation depths, and the next visit of the frame skips the splats no tile will
        # reach (depth-bound culling, csrc/fused.hip); a gated fallback or the speculative replay keeps the result exact.  Every
        # `cull_refresh`-th visit of a frame runs unculled (fresh bounds from the complete lists, fresh full table size, fresh schedule).
        self.cull_refresh = 16
        # margin of the depth bounds (percent of the splats walked beyond a tile's saturation point), per frame.  A fallback costs a whole
        # second binning + blend (~0.4 ms at 3 M @1080p), a wider margin only a few more instances (~35 us per million): measured over 40
        # training steps of the bench scene, margin 50 % -> 12 fallbacks, 1.093 ms/step; 100 % -> 1 fallback, 0.986 ms; 200 % -> none,
        # 1.026 ms (profiles/r02_margin_ab.log).  Base 100 %; the margin of a frame doubles when its previous visit fell back and decays
        # back after clean visits.  margin_fixed = <percent> pins it (tests / tools).
        self.margin_fixed = 0
        self.margin_lo, self.margin_hi = 100, 400
        self.frames = [_FrameState(self.margin_lo) for _ in range(n_frames)]
        self.sched = None
        self.tile_order = None
        self.fallbacks = 0                 # visits that were re-run unculled (observed one visit later)
        self.truncated_visits = 0          # unculled visits whose table turned out too short (observed one visit later)
        self.keep_size_predictions = True   # across a densification / re-sort (parameters_replaced): the reference's protocol, see there
        self.last_cull = False
        # fuse_optimizer: backward stops after the blend backward; FusedAdam.step() then runs the per-Gaussian backward fused
        # with the Adam update (csrc/fused.hip: project_backward_adam_kernel) -- parameter gradients never go to HBM.
        # Only valid when nothing needs the gradients between backward and the optimizer step (no gradient-hook DP exchange).
        self.fuse_optimizer = False
        # every consumer of this render's gradient records folds gradient replicas: true for the fused backward + Adam and for the
        # data-parallel moment compaction (csrc/dp.hip); the trainer clears it for anything else that reads the records
        self.fold_only_consumer = True
        self.pending = None
        self.probe_events = None      # measurement hook (bench.py): a list that receives an event pair around every blend backward launch
        # data-parallel hook (dp.MomentExchange.begin): called with (visible_chunkid, visible_chunks_num) as soon as the culling is
        # enqueued, so that the union-of-visibility collective runs on RCCL's stream underneath the whole forward + blend backward
        self.after_cull = None
        self._cull_scratch, self._cull_chunks, self._cull_epoch = None, -1, 0
        self.hot_counter = None       # device int32[1]: replica line counter (persistent; reset by the step's last kernel)
        # speculative culling (csrc/fused.hip): owned here, driven by FrameTrainer, which replays failed steps
        self.spec_poison = None       # device int32[1], sticky
        self.spec_words = None        # HostWords(2): [0] mirror of the poison word, [1] step number of the last fused Adam launch that ran
        self.spec_step = 0            # number of the training step being enqueued
        self.spec_forward = True      # False: culled forwards keep the gated repeat even while the steps are speculative (trainer.py: data-parallel
                                      # runs whose bounds fail too often -- there a failure costs every rank a replay, the repeat only 40 us)
        self.force_full = False       # the next render runs unculled (the first replayed step)
        self._debug_words = None      # HostWords(8), validate_tables only
        self.emission_mismatches = 0  # validate_tables: slots whose emission walk disagreed with the projection's tile count
        self._closed = False

    # -- compatibility views of the per-frame records (tests, tools, bench.py read them) --------------------------------------------
    @property
    def margin(self):
        return [f.margin for f in self.frames]

    @property
    def visits(self):
        return [f.visits for f in self.frames]

    @property
    def full_total(self):
        return [f.full_total for f in self.frames]

    @property
    def sched_cur(self):
        return [f.sched_cur for f in self.frames]

    # -- lifetime ---------------------------------------------------------------------------------------------------------------------
    def close(self, sync: bool = True):
        """Everything enqueued so far has finished before the renderer's device words and pinned words are released.  (The pinned words
        would survive anyway -- the arena quarantines them -- but the device tensors go back to torch's allocator.)
        sync=False (the finaliser): no device synchronisation from a garbage collection -- the pinned words go to the arena's quarantine,
        which re-issues them only behind a device synchronisation of its own, and the device tensors are stream-ordered in torch's allocator."""
        if self._closed:
            return
        self._closed = True
        if sync:
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:
                pass
        self.pending = None
        for w in (self._words, self.spec_words, self._debug_words):
            if w is not None:
                w.close()
        self.fb_vis = self.fb_total = self.fb_full = None

    def __del__(self):
        try:
            self.close(sync=False)
        except Exception:
            pass

    # -- speculation ------------------------------------------------------------------------------------------------------------------
    def enable_speculation(self, device):
        if self.spec_poison is None:
            self.spec_poison = torch.zeros((1,), dtype=torch.int32, device=device)
            self.spec_words = HostWords(2)
        return self.spec_words

    def disable_speculation(self):
        """back to the gated repeat.  The words are kept (a launch in flight may still store into them); only the mode changes."""
        self.spec_step = 0

    @property
    def speculating(self) -> bool:
        return self.spec_poison is not None and self.spec_step > 0

    def poisoned(self) -> bool:
        return self.spec_words is not None and int(self.spec_words.a[0]) != 0

    def applied_step(self) -> int:
        return int(self.spec_words.a[1]) if self.spec_words is not None else 0

    def clear_poison(self):
        self.spec_poison.zero_()
        self.spec_words.a[0] = 0

    def context(self, depth_order: int, replicas: bool, margin: int, speculate: bool) -> LgFusedCtx:
        """the LgFusedCtx of one call; `speculate`: hand the poison words over (a culled render that a fused Adam step follows, and that step)"""
        c = LgFusedCtx()
        c.struct_bytes = ctypes.sizeof(LgFusedCtx)
        c.depth_order = depth_order
        c.bound_margin_pct = max(1, int(margin))
        c.tile_scatter = 1 if self.tile_scatter else 0
        c.grad_replicas = 1 if replicas else 0
        c.hot_counter = self.hot_counter.data_ptr() if self.hot_counter is not None else None
        if speculate and self.speculating:
            c.poison = self.spec_poison.data_ptr()
            c.poison_host = self.spec_words.addr(0)
            c.applied_host = self.spec_words.addr(1)
            c.step_id = int(self.spec_step)
        if self.validate_tables:
            if self._debug_words is None:
                self._debug_words = HostWords(48)
            c.debug_validate = 1
            c.debug_words = self._debug_words.addr(0)
        return c

    def check_tables(self):
        """validate_tables: raise if a table check on the device found garbage since the last call (the run itself was kept alive)"""
        if self._debug_words is not None and int(self._debug_words.a[5]) != 0:
            # the key emission walked a splat to another tile count than the projection counted for it: padded / dropped on the device (the
            # table stays fully written), reported here because the two are meant to be the same function of the same floats
            import sys
            n, d = int(self._debug_words.a[5]), int(self._debug_words.a[6])
            self._debug_words.a[5] = 0
            self.emission_mismatches += n
            print(f"[litegs_amd validate] key emission: {n} slot(s) walked to a different tile count than the prefix sums hold "
                  f"(last: walked - counted = {d})", file=sys.stderr, flush=True)
        if self._debug_words is not None and (int(self._debug_words.a[8]) != 0 or int(self._debug_words.a[16]) != 0 or int(self._debug_words.a[32]) != 0):
            import sys
            d = [int(x) for x in self._debug_words.a]
            if d[32]:
                print(f"[litegs_amd validate] big-splat queue after dup_small: {'sub-queue over capacity' if d[33] == 1 else 'entry names no slot'}: sub-queue {d[34]} position {d[35]} "
                      f"entry 0x{d[36] & 0xffffffff:08x} sub-queue length {d[37]} capacity {d[38]} N {d[39]}", file=sys.stderr, flush=True)
                self._debug_words.a[32] = 0
            if d[8]:
                print(f"[litegs_amd validate] key emission: big-splat queue entry that names no slot: sub-queue {d[9]} position {d[10]} entry 0x{d[11] & 0xffffffff:08x} "
                      f"(sub-queue length {d[12]}, all queues {d[13]}, capacity per sub-queue {d[14]}, N {d[15]})", file=sys.stderr, flush=True)
            if d[16]:
                print(f"[litegs_amd validate] key emission (big splats): key {d[29]} out of range: splat {d[17]} part {d[18]} output {d[19]} owner rank {d[20]} slice {d[21]} "
                      f"(slice offset {d[22]}, first tile {d[23]}) run {d[24]} share {d[25]} table offset {d[26]} slices {d[27]} non-empty {d[28]} before {d[30]} isY {d[31]}",
                      file=sys.stderr, flush=True)
            self._debug_words.a[8] = 0; self._debug_words.a[16] = 0
            raise RuntimeError("litegs_amd: table validator: the key emission met garbage (details above)")
        if self._debug_words is not None and int(self._debug_words.a[0]) != 0:
            rec = [int(x) for x in self._debug_words.a]
            self._debug_words.a[0] = 0
            what = {1: "tile key out of range in the emitted table", 2: "tile range ends beyond the valid entries", 3: "splat id out of range in the grouped table",
                    4: "visible chunk id out of range", 5: "visible chunk count out of range",
                    6: "tile key out of range in the SORTED table", 7: "radix digit totals of the emission do not add up to the table length",
                    8: "a scratch word that must be zero on entry is not (bound: 1 = workspace-1 scratch after the projection, 2 = tile sort look-back table, 3 = tile sort tickets)",
                    10: "depth-sorted splat id out of range (hole in the splat sort's output)", 11: "prefix sums decrease (value < bound)"}
            raise RuntimeError(f"litegs_amd: table validator: {what.get(rec[0], 'code %d' % rec[0])}: where={rec[1]} value={rec[2]} bound={rec[3]} "
                               f"valid_entries={rec[4]} reports_so_far={rec[7]} emission_count_mismatches={rec[5]} (last walked-counted={rec[6]})")

    @staticmethod
    def sanitised_counts(reset: bool = True) -> dict:
        """Always-on counters of table words a kernel had to neutralise instead of indexing with them (csrc/lg_sanity.h): all zero for
        correct tables.  Blocking -- call at a synchronisation point (FrameTrainer.flush() does).  `truncated_tables` is informational."""
        out = (ctypes.c_int * 8)()
        check(lib().lg_sanitised_counts(out, 1 if reset else 0), "lg_sanitised_counts")
        names = ("emission_key", "emission_count_mismatch", "tile_range_key", "tile_scatter_key", "radix_scatter_index", "tilesort_id", "truncated_tables", "queue_entry")
        return {n: int(out[i]) for i, n in enumerate(names)}

    def note_fallback(self, k: int):
        """frame k was re-run unculled (gated repeat observed, or a speculative step replayed): widen its margin"""
        F = self.frames[k]
        self.fallbacks += 1
        F.clean_visits = 0
        if not self.margin_fixed:
            F.margin = min(F.margin * 2, self.margin_hi)

    def reset_feedback(self):
        """parameters were replaced / re-sorted: forget everything predicted from earlier visits (sizes, schedules, depth bounds)"""
        self.fb_vis[:] = 0; self.fb_total[:] = 0; self.fb_full[:] = 0
        for f in self.frames:
            f.reset(self.margin_fixed or self.margin_lo)

    def parameters_replaced(self, growth: float = 1.0):
        """Density control / a Morton re-sort replaced the parameters (`growth` = new / old point count).  Depth bounds and tile schedules
        describe the old cloud: dropped.

        keep_size_predictions = True (default) keeps the SIZE predictions of the frames in use, scaled by the growth, as the reference does
        (it never resets its feedback buffers, litegs/data.py:236-241): a densification adds a few percent of points, inside the 1.2x / 1.5x
        allocation margins, and an under-predicted table is truncated (GR/binning.cu:63), noticed, and sized exactly on the frame's next
        visit; frames out of use (evaluation frames) still start over.  Measured at 3 M / 150 cameras: epochs right after a densification
        cost 3.9-4.3 ms per iteration with the reset against 3.3-3.5 for the others (profiles/r04_convergence_3m_runs_11_13.md), the whole
        run 3.21 against 3.24-3.30 ms per iteration.  False: every frame's next visit is an exact, blocking first visit
        (GR/compact.cu:543-546, GR/binning.cu:152-163).  (Round 4 shipped False: the three long runs that had it on lost their second
        trainer to a memory access fault.  The fault had nothing to do with sizes -- profiles/r05_fault_root_cause.md: a negative tile-slice
        count in the key emission of needle-like splats -- and is fixed.)"""
        g = max(1.0, float(growth))
        for k, f in enumerate(self.frames):
            # a densification is TWO replacements with no visit in between (density control at the end of an epoch, the Morton re-sort at
            # the start of the next): a frame is out of use when it sat out two replacements in a row
            f.idle_replacements = 0 if f.visits > f.visits_at_replace else f.idle_replacements + 1
            keep = self.keep_size_predictions and f.idle_replacements < 2
            f.reset(self.margin_fixed or self.margin_lo)
            f.visits_at_replace = f.visits
            if keep:
                self.fb_vis[k] = int(np.ceil(g * int(self.fb_vis[k])))
                self.fb_total[k] = int(min(np.ceil(g * int(self.fb_total[k])), 2**31 - 1))
            else:
                self.fb_vis[k] = 0; self.fb_total[k] = 0
            self.fb_full[k] = 0

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
        F = R.frames[k]
        s = _s()
        needs_grad = any(ctx.needs_input_grad)
        visibility = _empty((chunks,), torch.bool, dev)
        vis_num = _empty((1,), torch.int32, dev)
        vis_ids = _empty((chunks,), torch.int64, dev)
        fb_vis_ptr = R._words.addr(k)
        fb_tot_ptr = R._words.addr(R.n_frames + k)
        fb_full_ptr = R._words.addr(2 * R.n_frames + k)
        common = (origin.data_ptr(), extend.data_ptr(), frame.planes.data_ptr(), chunks, frame.view_ptr, frame.proj_ptr, R.H, R.W, R.TH, R.TW,
                  int(degree), xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), sh_0.data_ptr(), sh_rest.data_ptr(), opacity.data_ptr(), S)
        stat = STATS.active
        pred_vis = int(R.fb_vis[k])
        pred_total = int(R.fb_total[k])
        # frames with long lists (previous visit): the splat sort + stable tile radix sort builds them, and gradient replicas are off --
        # with 1450 instances per tile the blend backward gains nothing from them (1196 us either way) while assigning the lines costs the
        # projection 49 us (one counter, thousands of atomics) and folding them the backward + Adam 134 us (profiles/r04_replicas_training_state.md)
        long_lists = R.long_list_global > 0 and pred_total > R.long_list_global * R.ntiles
        depth_order = R.depth_order
        if long_lists and depth_order == 2:
            depth_order = 0
        replicas = bool(R.replicas_enabled and R.fuse_optimizer and R.fold_only_consumer and not stat and needs_grad and not long_lists)
        if replicas and R.hot_counter is None:
            R.hot_counter = _empty((1,), torch.int32, dev, zero=True)
        do_cull = 1
        if pred_vis <= 0:                                    # first visit: blocking count (GR/compact.cu:543-546)
            c0 = R.context(depth_order, False, F.margin, False)
            check(L.lg_fused_stage1(ctypes.byref(c0), *common, 1, visibility.data_ptr(), vis_num.data_ptr(), vis_ids.data_ptr(), 0, None, 0,
                                    fb_vis_ptr, None, *R.cull_scratch(chunks, dev), None, None, s),
                  "fused cull")
            A = int(vis_num.item())
            do_cull = 0
        else:
            A = min(int(1.2 * pred_vis), chunks)
        A = max(A, 1)
        N = A * S
        ws1_bytes = L.lg_fused_workspace1_bytes(N)
        ws1 = _empty((ws1_bytes,), torch.uint8, dev, align=64)            # 64-byte records read by 64-byte scalar loads
        # The reference rasterises along the statistics helper's cached heavy-first tile list whenever a frame has one, i.e. in every render
        # after the first statistics epoch (litegs/render/__init__.py:75-79), and so does the executor by default: from then on its own
        # schedule, depth bounds and speculative culling are idle.  The list is a permutation of ALL tiles -- a schedule, not a selection --
        # so stat_schedule_always = False keeps the executor's own machinery outside statistics renders (same image, test_gpu_stats.py).
        tiles = STATS.schedule_for_current_frame() if (stat or R.stat_schedule_always) else None
        if tiles is not None and (tiles.shape[1] != R.ntiles or tiles.device != dev):
            # the statistics helper is a process-wide singleton keyed by frame index (as the reference's, statistic_helper.py:14): a list
            # cached by another trainer at another resolution is not this frame's schedule.  (A foreign list of the right length is a
            # permutation of all tiles -- a schedule, harmless.)
            tiles = None
        # depth-bound culling: bookkeeping of the sizing feedback (the emitted total of a culled visit is not the full table length)
        if F.last_unculled and pred_total > F.last_capacity > 0:
            # the previous (unculled) visit needed more entries than its predicted table held: its tail was dropped, as in the reference
            # (GR/binning.cu:63, silent there).  Counted here, and this visit sizes its table exactly (the blocking first-visit path).
            R.truncated_visits += 1
            pred_total = 0
            R.fb_total[k] = 0
        if F.last_unculled and pred_total > 0:
            F.full_total = pred_total                         # the previous visit of this frame emitted everything
        fb_full = int(R.fb_full[k])
        if fb_full > 0:                                       # ... or a fallback re-ran it in full
            if fb_full > F.last_capacity > 0:                 # ... into a table that was too short for it: same treatment
                R.truncated_visits += 1
                pred_total = 0
                R.fb_total[k] = 0
            F.full_total = max(F.full_total, fb_full)
            R.fb_full[k] = 0
            R.note_fallback(k)
        elif not R.margin_fixed:
            F.clean_visits += 1
            if F.clean_visits >= 6 and F.margin > R.margin_lo:
                F.margin = max(R.margin_lo, (F.margin * 3) // 4)
                F.clean_visits = 0
        use_sched = not stat and tiles is None
        if use_sched and R.sched is None:
            R.sched = _empty((R.n_frames, 2, L.lg_sched_words(R.H, R.W, R.TH, R.TW)), torch.int32, dev)
            R.tile_order = _empty((R.n_frames, R.ntiles), torch.int32, dev)
        in_ptr = out_ptr = None
        if use_sched:
            if F.sched_valid:
                in_ptr = R.sched[k, F.sched_cur].data_ptr()
            out_ptr = R.sched[k, 1 - F.sched_cur].data_ptr()
        refresh = F.visits % R.cull_refresh == 0
        cull = bool(R.cull_enabled and in_ptr is not None and F.full_total > 0 and pred_total > 0 and not refresh and not R.force_full)
        R.force_full = False
        order_ptr = (R.tile_order.data_ptr() + 4 * R.ntiles * k) if use_sched else None
        order_in = order_ptr if (use_sched and F.order_valid) else None
        order_out = order_ptr if (use_sched and (refresh or not F.order_valid)) else None
        F.visits += 1
        # a culled render that a fused Adam step follows may run speculatively (no gated repeat); anything else keeps the repeat
        cx = R.context(depth_order, replicas, F.margin, cull and R.fuse_optimizer and needs_grad and R.spec_forward)
        check(L.lg_fused_stage1(ctypes.byref(cx), *common, do_cull, visibility.data_ptr(), vis_num.data_ptr(), vis_ids.data_ptr(), A,
                                ws1.data_ptr(), ws1_bytes, fb_vis_ptr if do_cull else None, fb_tot_ptr,
                                *(R.cull_scratch(chunks, dev) if do_cull else (None, 0)), in_ptr if cull else None, out_ptr, s), "fused stage1")
        if R.after_cull is not None and needs_grad:
            R.after_cull(vis_ids, vis_num)
        if pred_total <= 0:                                  # first visit: blocking table size (GR/binning.cu:152-163)
            off = L.lg_fused_total_offset(N)
            table_len = int(ws1[off:off + 4].view(torch.int32).item())
        elif cull:
            table_len = int(1.5 * F.full_total)              # capacity for the fallback's full table
        else:
            table_len = int(1.5 * max(pred_total, F.full_total))
        table_len = max(table_len, 1)
        # the emitted total was predicted under the margin of the visit before last: scale the culled table when the bounds got wider
        grow = max(1.0, F.margin_written / max(F.margin_emitted, 1))
        len_cull = min(table_len, int(1.5 * grow * pred_total) + 65536) if cull else table_len
        ws2_bytes = L.lg_fused_workspace2_bytes(table_len, N, R.H, R.W, R.TH, R.TW)
        ws2 = _empty((ws2_bytes,), torch.uint8, dev)
        img = _empty((1, 3, R.Hp, R.Wp), torch.float32, dev)
        trans = _empty((1, 1, R.Hp, R.Wp), torch.float32, dev)
        last = _empty((1, 1, R.Hp, R.Wp), torch.int16, dev)
        K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
        fc = fw = None
        # statistic epochs, 8x16 tiles: fragment count / weight / err_square travel in slots 9-11 of the blend backward's gradient record
        # (csrc/raster.hip, STAT == 2: no per-splat atomics of their own); other tile shapes keep the forward's two counter arrays
        stat_in_record = stat and bool(L.lg_stat_in_record_supported(R.TH, R.TW))
        if stat:
            if not stat_in_record:
                fc = _empty((1, 1, N), torch.int32, dev, zero=True)
                fw = _empty((1, 1, N), torch.float32, dev, zero=True)
            STATS.set_compaction(vis_ids[:A], vis_num)
            if not (stat_in_record and needs_grad):          # (with the statistics in the record the backward counts the visible ones too)
                a_off = L.lg_fused_alloc_offset(N)            # b_visible = allocate_size != 0 (wrapper.py:733-736)
                STATS.add_visible((ws1[a_off:a_off + 4 * N].view(torch.int32) != 0).view(1, N))
        # gradient accumulator of the blend backward: allocated here so that stage 2 can clear it on the side (no memset launch later)
        pg_lines = L.lg_fused_grad_lines(N) if replicas else N
        pg = _empty((pg_lines, L.lg_packed_grad_floats()), torch.float32, dev) if needs_grad else None
        check(L.lg_fused_stage2(ctypes.byref(cx), A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes, tp, K,
                                1 if stat else 0, img.data_ptr(), trans.data_ptr(), last.data_ptr(),
                                fc.data_ptr() if fc is not None else None, fw.data_ptr() if fw is not None else None,
                                pg.data_ptr() if pg is not None else None,
                                order_in, order_out, in_ptr, out_ptr, 1 if cull else 0, len_cull, fb_full_ptr,
                                frame.view_ptr, frame.proj_ptr, int(degree), chunks,
                                xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), sh_0.data_ptr(), sh_rest.data_ptr(), opacity.data_ptr(),
                                vis_ids.data_ptr(), vis_num.data_ptr(), s), "fused stage2")
        if use_sched:
            F.sched_cur = 1 - F.sched_cur
            F.sched_valid = True
        F.last_unculled = not cull
        F.last_capacity = table_len
        R.last_cull = cull
        if cull:
            F.margin_emitted = F.margin_written
        F.margin_written = F.margin
        R.last_ws1 = (ws1, N)                              # for tests: the fallback flag lives in workspace 1 (lg_fused_flags_offset)
        R.last_ws2 = (ws2, table_len, N)                   # for tools: the tile range table lives in workspace 2 (lg_fused_tile_start_offset)
        R.last_ctx = cx                                    # ... at an offset that depends on the frame's depth-order mode
        if order_out is not None:
            F.order_valid = True
        ctx.order_ptr = order_ptr if (use_sched and F.order_valid) else None
        ctx.pg = pg
        ctx.replicas = replicas
        # the backward finds its lists where THIS frame's forward put them: same depth-order mode, same replica setting, no speculation
        ctx.cx = R.context(depth_order, replicas, F.margin, False)
        if stat or (tiles is not None and R.refresh_stat_schedule):
            STATS.update_tile_schedule(last, R.TH, R.TW)
        ctx.R, ctx.frame, ctx.meta = R, frame, (A, S, table_len, int(degree), chunks, sh_rest.shape[0], ws1_bytes, ws2_bytes, stat)
        ctx.tiles = tiles
        ctx.stat_bufs = (fc, fw, stat_in_record)
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
        cx = ctx.cx
        fc, fw, stat_in_record = ctx.stat_bufs
        esq = _empty((1, 1, N), torch.float32, dev, zero=True) if (stat and not stat_in_record) else None
        tiles = ctx.tiles
        K, tp = (tiles.shape[1], tiles.data_ptr()) if tiles is not None else (0, None)
        if R.fuse_optimizer:
            if R.probe_events is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            check(L.lg_fused_backward(ctypes.byref(cx), A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes,
                                      frame.view_ptr, frame.proj_ptr, degree, chunks, Rr, vis_ids.data_ptr(), vis_num.data_ptr(),
                                      xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), opacity.data_ptr(), tp, K,
                                      trans.data_ptr(), last.data_ptr(), g_img.data_ptr(), None, None, 1 if stat else 0,
                                      pg.data_ptr(), pg_zero, esq.data_ptr() if esq is not None else None, None, None, None, None, None, None, ctx.order_ptr, _s()),
                  "fused blend backward")
            if R.probe_events is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                R.probe_events.append((ev0, ev1))
            if stat and stat_in_record:
                STATS.set_compaction(vis_ids[:A], vis_num)
                STATS.accumulate_records(pg, ws1.data_ptr() + L.lg_fused_packed_offset(N), ws1.data_ptr() + L.lg_fused_alloc_offset(N), A)
            elif stat:
                STATS.add_moments("fragment_weight", fw, fw * fw, fc)
                STATS.add_moments("fragment_err", _d_opacity(pg, ws1, N), esq, fc)
            # ws1 rides along: its tile counts tell the fused backward + Adam which gradient records can only be zero
            R.pending = dict(pg=pg, A=A, S=S, frame=frame, degree=degree, chunks=chunks, Rr=Rr, vis_ids=vis_ids, vis_num=vis_num, ws1=ws1,
                             replicas=ctx.replicas)
            if ctx.replicas:                              # for consumers other than FusedAdam (the data-parallel compaction folds them too)
                R.pending["hot_of"] = ws1.data_ptr() + L.lg_fused_hot_offset(N)
                R.pending["hot_counter"] = R.hot_counter.data_ptr()
            return (None,) * 11
        d_pos = _empty((3, A, S), torch.float32, dev)
        d_scale = _empty((3, A, S), torch.float32, dev)
        d_rot = _empty((4, A, S), torch.float32, dev)
        d_sh0 = _empty((3, A, S), torch.float32, dev)
        d_shr = _empty((Rr * 3, A, S), torch.float32, dev)
        d_opa = _empty((1, A, S), torch.float32, dev)
        check(L.lg_fused_backward(ctypes.byref(cx), A, S, table_len, R.H, R.W, R.TH, R.TW, ws1.data_ptr(), ws1_bytes, ws2.data_ptr(), ws2_bytes,
                                  frame.view_ptr, frame.proj_ptr, degree, chunks, Rr, vis_ids.data_ptr(), vis_num.data_ptr(),
                                  xyz.data_ptr(), scale.data_ptr(), rot.data_ptr(), opacity.data_ptr(), tp, K,
                                  trans.data_ptr(), last.data_ptr(), g_img.data_ptr(), None, None, 1 if stat else 0,
                                  pg.data_ptr(), pg_zero, esq.data_ptr() if esq is not None else None,
                                  d_pos.data_ptr(), d_scale.data_ptr(), d_rot.data_ptr(), d_sh0.data_ptr(), d_shr.data_ptr(), d_opa.data_ptr(),
                                  ctx.order_ptr, _s()),
              "fused backward")
        if stat and stat_in_record:
            STATS.set_compaction(vis_ids[:A], vis_num)
            STATS.accumulate_records(pg, ws1.data_ptr() + L.lg_fused_packed_offset(N), ws1.data_ptr() + L.lg_fused_alloc_offset(N), A)
        elif stat:
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
        self.skip_untouched = True           # exact skip of no-op updates (csrc/fused.hip); False: every visible Gaussian is read and written

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
        hot = None
        if pend.get("replicas") and "ws1" in pend:
            hot = pend["ws1"].data_ptr() + lib().lg_fused_hot_offset(pend["A"] * pend["S"])
        cx = R.context(R.depth_order, bool(pend.get("replicas")), 100, True)      # speculative mode: this launch honours / reports the poison word
        check(lib().lg_fused_backward_adam(ctypes.byref(cx), hot, pend["A"], pend["S"], R.H, R.W, fr.view_ptr, fr.proj_ptr, pend["degree"], pend["chunks"], pend["Rr"],
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
        step_id = 0
        if R.speculating and getattr(exchange, "spec", None) is not None:       # rank-consistent speculative culling (litegs_amd/dp.py)
            step_id = int(R.spec_step)
        return exchange.step(pend, cams, ps, ms, vs, lr6, float(self.groups[0]["eps"]), R.H, R.W, slot,
                             self._touched_flags() if self.skip_untouched else None, step_id=step_id)

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
