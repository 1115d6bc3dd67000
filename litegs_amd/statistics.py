"""Per-Gaussian statistics side channel (densification inputs) -- host-side mirror of
``litegs/utils/statistic_helper.py`` (StatisticsHelper :14-245, StatisticGuard :247-260).

The object is algorithmic state, not logging: while ``active`` the raster kernels run in statistic mode
(fragment counts / weights / error moments) and the per-frame "heavy tiles first" schedule is refreshed.
All accumulation into the full [*, chunks, S] tensors goes through the GPU-driven sparse scatter
(``gpu_driven_pipeline_sparse_op``), because the tail of every compacted tensor is dirty by design.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .binding import ops as fused


class _Moments:
    def __init__(self, lead_shape, chunks, S, device):
        self.sum = torch.zeros((*lead_shape, chunks, S), device=device)
        self.square_sum = torch.zeros((*lead_shape, chunks, S), device=device)
        self.count = torch.zeros((chunks, S), device=device, dtype=torch.int32)


class Statistics:
    def __init__(self):
        self.active = False
        self.enabled_for_epoch: Callable[[int], bool] = lambda epoch: False
        self.chunks = 0
        self.S = 0
        self.device = None
        self.moments: Dict[str, _Moments] = {}
        self.visible_count: Optional[torch.Tensor] = None
        self.compact_ids: Optional[torch.Tensor] = None
        self.compact_count: Optional[torch.Tensor] = None
        self.tile_schedule: Dict[object, torch.Tensor] = {}     # frame key -> int32[tiles] (1-based ids, heavy first)
        self.tile_blend_count: Dict[object, torch.Tensor] = {}
        self.current_frame = None

    # -- lifecycle ---------------------------------------------------------------------------------
    def reset(self, chunks: int, S: int, enabled_for_epoch: Optional[Callable[[int], bool]] = None, device="cuda"):
        self.active = False
        if enabled_for_epoch is not None:
            self.enabled_for_epoch = enabled_for_epoch
        self.chunks, self.S, self.device = chunks, S, device
        self.moments = {}
        self.visible_count = torch.zeros((chunks, S), dtype=torch.int32, device=device)
        self.compact_ids = None
        self.compact_count = None

    def epoch(self, epoch: int) -> "_Guard":
        return _Guard(self if self.enabled_for_epoch(epoch) else None)

    # -- hooks called from render ------------------------------------------------------------------
    def set_compaction(self, visible_chunkid: torch.Tensor, visible_chunks_num: torch.Tensor):
        self.compact_ids, self.compact_count = visible_chunkid, visible_chunks_num

    @torch.no_grad()
    def add_visible(self, visible_mask: torch.Tensor):
        """visible_mask bool[V, A*S] over the compacted Gaussians."""
        add = visible_mask.sum(0, dtype=torch.int32).reshape(1, -1, self.S)
        fused.gpu_driven_pipeline_sparse_op(self.visible_count.view(1, self.chunks, self.S), add, self.compact_ids, self.compact_count, "add")

    @torch.no_grad()
    def add_moments(self, key: str, value_sum: torch.Tensor, square_sum: torch.Tensor, count: torch.Tensor):
        """value_sum / square_sum [..., A*S] float32, count int32[..., A*S] (compacted)."""
        value_sum = value_sum.reshape(*value_sum.shape[:-1], -1, self.S)
        square_sum = square_sum.reshape(*square_sum.shape[:-1], -1, self.S)
        mom = self.moments.get(key)
        if mom is None:
            mom = _Moments(value_sum.shape[:-2], self.chunks, self.S, value_sum.device)
            self.moments[key] = mom
        A = value_sum.shape[-2]
        fused.gpu_driven_pipeline_sparse_op(mom.sum.view(-1, self.chunks, self.S), value_sum.reshape(-1, A, self.S).contiguous(),
                                            self.compact_ids, self.compact_count, "add")
        fused.gpu_driven_pipeline_sparse_op(mom.square_sum.view(-1, self.chunks, self.S), square_sum.reshape(-1, A, self.S).contiguous(),
                                            self.compact_ids, self.compact_count, "add")
        fused.gpu_driven_pipeline_sparse_op(mom.count.view(-1, self.chunks, self.S), count.reshape(-1, A, self.S).contiguous(),
                                            self.compact_ids, self.compact_count, "add")

    @torch.no_grad()
    def accumulate_records(self, packed_grad: torch.Tensor, packed_ptr: int, alloc_ptr: int, A: int):
        """native executor, statistic epochs: visible_count and the moments "fragment_weight" / "fragment_err" of one frame in one pass over
        its gradient records (csrc/compact.hip stat_accumulate_kernel) -- what add_visible + two add_moments calls accumulate, without the
        dozen elementwise / scatter launches in between.  set_compaction() must have been called for the frame."""
        from ._lib import check, lib
        for key in ("fragment_weight", "fragment_err"):
            if key not in self.moments:
                self.moments[key] = _Moments((1, 1), self.chunks, self.S, packed_grad.device)
        w, e = self.moments["fragment_weight"], self.moments["fragment_err"]
        check(lib().lg_stat_accumulate(packed_grad.data_ptr(), packed_ptr, alloc_ptr, self.compact_ids.data_ptr(), self.compact_count.data_ptr(),
                                       int(A), self.chunks, self.S, self.visible_count.data_ptr(),
                                       w.sum.data_ptr(), w.square_sum.data_ptr(), w.count.data_ptr(),
                                       e.sum.data_ptr(), e.square_sum.data_ptr(), e.count.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "stat_accumulate")

    @torch.no_grad()
    def update_tile_schedule(self, last_contributor: torch.Tensor, th: int, tw: int):
        """Per-tile max blend depth -> heavy-first tile order for the next time this frame is rendered
        (statistic_helper.py:68-79)."""
        if self.current_frame is None:
            return
        N, _, Hp, Wp = last_contributor.shape
        ty, tx = Hp // th, Wp // tw
        if last_contributor.is_cuda and N == 1 and last_contributor.dtype == torch.int16 and last_contributor.is_contiguous():
            # the executor's own kernels (csrc/raster.hip): per-tile maximum of last_contributor by one wave per tile, heavy-first order by
            # a counting sort on min(count, 1023) -- two launches instead of the reshape / permute / max / sort chain of torch kernels
            # (a dozen launches per step of a statistics epoch, profiles/r05_training_state_timeline.md).  Ties, and tiles beyond 1023
            # splats among themselves, come in tile order; torch.sort's order of equal counts is unspecified anyway.
            from ._lib import check, lib
            ntiles = ty * tx
            work = torch.empty((1, ntiles + 1), dtype=torch.int32, device=last_contributor.device)
            order = torch.empty((1, ntiles), dtype=torch.int32, device=last_contributor.device)
            s = torch.cuda.current_stream().cuda_stream
            check(lib().lg_tile_work_from_last(last_contributor.data_ptr(), 1, Hp, Wp, th, tw, work.data_ptr(), s), "tile_work_from_last")
            check(lib().lg_tile_order(work.data_ptr(), 1, ntiles, order.data_ptr(), s), "tile_order")
            self.tile_blend_count[self.current_frame] = work[0, 1:]
            self.tile_schedule[self.current_frame] = order[0]
            return
        per_tile = last_contributor.reshape(N, ty, th, tx, tw).permute(1, 3, 0, 2, 4).reshape(ty * tx, -1).max(dim=1).values
        self.tile_blend_count[self.current_frame] = per_tile
        self.tile_schedule[self.current_frame] = (per_tile.sort(descending=True)[1].int() + 1)

    def schedule_for_current_frame(self) -> Optional[torch.Tensor]:
        t = self.tile_schedule.get(self.current_frame)
        return None if t is None else t.unsqueeze(0)

    # -- queries used by the density controller --------------------------------------------------------
    @torch.no_grad()
    def mean(self, key: str):
        mom = self.moments.get(key)
        if mom is None:
            return None
        m = mom.sum / (mom.count + 1e-9)
        return m.reshape(*m.shape[:-2], -1), mom.count.reshape(-1)

    @torch.no_grad()
    def var(self, key: str):
        mom = self.moments.get(key)
        if mom is None:
            return None
        mean = mom.sum / (mom.count + 1)
        sq = mom.square_sum / (mom.count + 1)
        v = (sq - mean ** 2).clamp_min(0)
        return v.reshape(*v.shape[:-2], -1), mom.count.reshape(-1)

    def never_visible(self) -> torch.Tensor:
        return (self.visible_count == 0).reshape(-1)

    # -- data parallelism ----------------------------------------------------------------------------
    @torch.no_grad()
    def all_reduce(self, group=None) -> None:
        """Sum the accumulators over the ranks of a data-parallel job (each rank saw different frames): call once,
        right before the density controller reads mean()/var()/never_visible(), so every rank takes identical prune/clone/split
        decisions.  Sums, squared sums and counts are additive, so the reduced moments equal those of a single process that had
        rendered all frames.  Keys are visited in sorted order (the same collective sequence on every rank)."""
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        if self.visible_count is not None:
            dist.all_reduce(self.visible_count, op=dist.ReduceOp.SUM, group=group)
        for key in sorted(self.moments):
            mom = self.moments[key]
            for t in (mom.sum, mom.square_sum, mom.count):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class _Guard:
    def __init__(self, stats: Optional[Statistics]):
        self.stats = stats

    def __enter__(self):
        if self.stats is not None:
            self.stats.active = True

    def __exit__(self, *exc):
        if self.stats is not None:
            self.stats.active = False


STATS = Statistics()
