"""One-camera-frame-per-GPU data parallelism over RCCL (the reference is single-GPU only).

Every rank holds a full replica of the Gaussians and the Adam state; per step rank r renders frame
``perm[step*W + r]`` and the parameter gradients are averaged across ranks before the (replicated) sparse
Adam step, so all replicas stay bit-identical.  The exchange:

 1. each rank marks its visible chunks in an int32[chunks] mask; ``all_reduce(MAX)`` gives the UNION of
    visibility (23 k entries at 3 M Gaussians: latency only);
 2. the union mask is compacted on the device (ordered, no host sync) into ``(union_ids, union_count)`` --
    Adam must touch exactly the chunks some rank saw (invisible chunks keep param/m/v untouched, as on 1 GPU);
 3. the six compact gradients are scatter-added into ONE persistent dense buffer ``[59, chunks, S]`` (rows:
    xyz 3, scale 3, rot 4, sh_0 3, sh_rest 45, opacity 1) and reduced with ONE ``all_reduce(AVG)`` --
    a single large collective (708 MB at 3 M) instead of six, because xGMI rings are per-link bound and
    small messages only add latency;
 4. each parameter's ``.grad`` becomes a dense view into that buffer; ``SparseGaussianAdam`` runs its
    dense-gradient kernel over ``union_ids``.

Gradient semantics: MEAN over ranks (keeps the single-GPU learning rates).  Statistics for densification are
not exchanged yet ("next").  The collective / bookkeeping logic is device agnostic: the three
primitive ops (mark, compact, scatter-add) come from an ``ops`` object -- ``HipOps`` (the HIP kernels; default)
-- so the N>1 logic is covered by world_size-2 gloo tests on CPU that inject a plain-torch ``ops``.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class HipOps:
    """Device primitives on the HIP kernels (no CPU path)."""

    @staticmethod
    def mark(mask: torch.Tensor, ids: torch.Tensor, count: torch.Tensor) -> None:
        from ._lib import check, lib
        check(lib().lg_mark_chunks(ids.data_ptr(), count.data_ptr(), ids.shape[0], mask.data_ptr(), torch.cuda.current_stream().cuda_stream), "mark_chunks")

    @staticmethod
    def compact(mask: torch.Tensor):
        from ._lib import check, lib
        M = mask.shape[0]
        count = torch.empty((1,), dtype=torch.int32, device=mask.device)
        ids = torch.empty((M,), dtype=torch.int64, device=mask.device)
        check(lib().lg_compact_mask(mask.data_ptr(), M, count.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream), "compact_mask")
        return ids, count

    @staticmethod
    def scatter_add(dense: torch.Tensor, compact: torch.Tensor, ids: torch.Tensor, count: torch.Tensor) -> None:
        from . import fused
        fused.gpu_driven_pipeline_sparse_op(dense, compact, ids, count, "add")


class GradientExchange:
    def __init__(self, params: Sequence[torch.Tensor], world: int, ops=HipOps, group=None):
        self.world, self.ops, self.group = world, ops, group
        p0 = params[0]
        self.chunks, self.S = p0.shape[-2], p0.shape[-1]
        self.rows = [int(p.numel() // (self.chunks * self.S)) for p in params]
        self.buf = torch.zeros((sum(self.rows), self.chunks, self.S), dtype=torch.float32, device=p0.device)
        self.mask = torch.zeros((self.chunks,), dtype=torch.int32, device=p0.device)
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.use_avg = backend == "nccl"           # RCCL implements AVG; gloo does not

    def hook(self, params: List[torch.Tensor], vis_id: torch.Tensor, vis_num: torch.Tensor):
        """Called between backward and the optimizer step.  Returns (union_ids, union_count)."""
        # 1. union of visibility
        self.mask.zero_()
        self.ops.mark(self.mask, vis_id, vis_num)
        dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group)
        # 2. ordered union list, on the device
        union_ids, union_count = self.ops.compact(self.mask)
        # 3. one dense buffer, one collective
        self.buf.zero_()
        row = 0
        for p, r in zip(params, self.rows):
            g = p.grad
            view = self.buf[row:row + r]
            if g is not None:
                if hasattr(g, "compacted_values"):
                    self.ops.scatter_add(view, g.compacted_values.reshape(r, -1, self.S), vis_id, vis_num)
                else:
                    view.add_(g.reshape(r, self.chunks, self.S))
            row += r
        if self.use_avg:
            dist.all_reduce(self.buf, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
            self.buf.mul_(1.0 / self.world)
        # 4. dense .grad views
        row = 0
        for p, r in zip(params, self.rows):
            p.grad = self.buf[row:row + r].view(p.shape)
            row += r
        return union_ids, union_count


def frame_for(step: int, rank: int, world: int, n_frames: int, perm=None) -> int:
    """Rank r trains frame perm[(step*world + r) mod n_frames]: disjoint frames within a step, same permutation everywhere."""
    k = (step * world + rank) % n_frames
    return int(perm[k]) if perm is not None else k
