"""One-camera-frame-per-GPU data parallelism over RCCL (the reference is single-GPU only).

Every rank holds a full replica of the Gaussians and the Adam state; per step rank r renders frame
``perm[step*W + r]`` and the parameter gradients are averaged across ranks before the (replicated) sparse
Adam step, so all replicas stay bit-identical.  The exchange:

 1. each rank marks its visible chunks in an int32[chunks] mask; ``all_reduce(MAX)`` gives the UNION of
    visibility (23 k entries at 3 M Gaussians: latency only);
 2. the union mask is compacted on the device (ordered, no host sync) into ``(union_ids, union_count)`` plus the
    inverse map chunk -> position in the union -- Adam must touch exactly the chunks some rank saw (invisible
    chunks keep param/m/v untouched, as on 1 GPU);
 3. the six compact gradients are scatter-added into ONE persistent buffer packed over the union,
    ``[59, U, S]`` (rows: xyz 3, scale 3, rot 4, sh_0 3, sh_rest 45, opacity 1), and reduced with ONE
    ``all_reduce(AVG)``: a single large collective instead of six (xGMI rings are per-link bound, small messages
    only add latency), and only the chunks somebody saw travel -- 38 / 53 / 69 % of the dense 708 MB at
    W = 2 / 4 / 8 for the 3 M benchmark scene;
 4. each parameter's ``.grad`` becomes a CompactedTensor over ``union_ids`` viewing that buffer; the sparse Adam
    kernel runs over the union list exactly as it runs over a rank's own visible list on one GPU.

The collective's size must be known on the host and be the same on every rank.  It follows the reference's GPU-driven
sizing protocol (litegs/data.py:236-241): a pinned per-slot feedback buffer receives the union count of step k and
sizes the buffer of the next visit of that slot (x1.2); only the first visit blocks on the count.  The count derives
from the all-reduced mask, so every rank reads the same value.  A union that outgrows the prediction loses its last
chunks for that step (silent truncation, as everywhere in this protocol).

That dense exchange (mode "dense") moves 2(W-1)/W x 360..665 MB per step: on xGMI, where two GPUs share ONE link (~77 GB/s per
direction), it costs several compute steps.  The default is therefore mode "sparse": a frame leaves a non-zero gradient only on
the Gaussians it actually blended (9 k of 3 M in the benchmark scene, where ~80 near splats saturate every tile; at most the
~0.35 M that touch a tile and survive occlusion), so every rank

 a. compacts the Gaussians whose 59 gradient values are not all zero (exact test: a row of zeros contributes nothing to a sum),
    -> [59 values + global index] x K: 2 MB (benchmark scene) .. 85 MB instead of 360..665 MB;
 b. learns the largest K of the job from the SAME small collective that builds the union of visible chunks (K rides as one more
    element of the MAX-reduced mask) -- one host read per step, which is affordable here: the step is communication bound and the
    host has nothing to enqueue ahead of the exchange anyway;
 c. ``all_gather``s the fixed-size [60, Kmax] blocks: they arrive over the W-1 direct links in parallel;
 d. adds the W blocks in RANK ORDER (one ``index_add_`` per rank: deterministic, the replicas stay bit-identical) into one dense
    gradient buffer, pre-scaled by 1/W; padding entries add 0.0 to element 0.  Adam then runs over the union of visible chunks
    with the dense-gradient kernel path.

Gradient semantics: MEAN over ranks (keeps the single-GPU learning rates).  Densification statistics are summed across
ranks by ``litegs_amd.statistics.Statistics.all_reduce`` right before the density controller reads them.  The collective / bookkeeping logic is device agnostic: the primitive ops
(mark, compact, scatter-add) come from an ``ops`` object -- ``HipOps`` (the HIP kernels; default) -- so the N>1
logic is covered by world_size-2 gloo tests on CPU that inject a plain-torch ``ops``.
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
        """-> (ids int64[M] ascending union then arange tail, count int32[1], rank int64[M]: position of chunk m in ids, for union chunks)"""
        from ._lib import check, lib
        M = mask.shape[0]
        count = torch.empty((1,), dtype=torch.int32, device=mask.device)
        ids = torch.empty((M,), dtype=torch.int64, device=mask.device)
        rank = torch.empty((M,), dtype=torch.int64, device=mask.device)
        check(lib().lg_compact_mask_rank(mask.data_ptr(), M, count.data_ptr(), ids.data_ptr(), rank.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "compact_mask")
        return ids, count, rank

    @staticmethod
    def scatter_add(dense: torch.Tensor, compact: torch.Tensor, ids: torch.Tensor, count: torch.Tensor) -> None:
        """dense[:, ids[a], :] += compact[:, a, :] for a < count; ids outside dense are dropped"""
        from . import fused
        fused.gpu_driven_pipeline_sparse_op(dense, compact, ids, count, "add")

    @staticmethod
    def feedback(host_slot: torch.Tensor, count: torch.Tensor) -> None:
        from ._lib import check, lib
        check(lib().lg_feedback_d2h(host_slot.data_ptr(), count.data_ptr(), torch.cuda.current_stream().cuda_stream), "feedback")


class GradientExchange:
    def __init__(self, params: Sequence[torch.Tensor], world: int, ops=HipOps, group=None, n_slots: int = 64, mode: str = None):
        import os
        self.mode = mode or os.environ.get("LITEGS_DP_EXCHANGE", "sparse")
        if self.mode not in ("sparse", "dense"):
            raise ValueError("GradientExchange mode must be 'sparse' or 'dense'")
        self.world, self.ops, self.group = world, ops, group
        p0 = params[0]
        self.chunks, self.S = p0.shape[-2], p0.shape[-1]
        self.rows = [int(p.numel() // (self.chunks * self.S)) for p in params]
        self.nrows = sum(self.rows)
        self.flat = torch.zeros((self.nrows * self.chunks * self.S,), dtype=torch.float32, device=p0.device)   # capacity: dense
        self.mask = torch.zeros((self.chunks + 1,), dtype=torch.int32, device=p0.device)        # [chunks] visibility | [1] nonzero count
        self.last_k = (0, 0)
        self._touched = None                       # columns of the dense gradient buffer the previous sparse exchange added to
        self.fb_union = torch.zeros((n_slots,), dtype=torch.int32)
        if p0.is_cuda:                             # the device stores into these words: library arena, never unmapped (hostwords.py)
            from .hostwords import pinned_int32
            self.fb_union, self._fb_union_owner = pinned_int32((n_slots,))
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.use_avg = backend == "nccl"           # RCCL implements AVG; gloo does not
        self.last_alloc = 0

    def rebind(self, params: Sequence[torch.Tensor]) -> None:
        """after density control / a re-sort replaced the parameters: new chunk count -> new buffers, size predictions dropped"""
        p0 = params[0]
        if p0.is_cuda:
            torch.cuda.current_stream().synchronize()            # feedback words may still be in flight
        self.fb_union.zero_()
        if self._touched is not None:
            self.flat.view(self.nrows, -1)[:, self._touched] = 0.0
            self._touched = None
        if p0.shape[-2] != self.chunks:
            self.chunks = p0.shape[-2]
            self.flat = torch.zeros((self.nrows * self.chunks * self.S,), dtype=torch.float32, device=p0.device)
            self.mask = torch.zeros((self.chunks + 1,), dtype=torch.int32, device=p0.device)

    def hook(self, params: List[torch.Tensor], vis_id: torch.Tensor, vis_num: torch.Tensor, slot: int = 0):
        """Called between backward and the optimizer step.  ``slot``: any integer that is the same on all ranks and recurs with the
        same set of frames (e.g. the step index modulo the steps per epoch).  Returns (union_ids[:U_alloc], union_count)."""
        from .wrapper import CompactedTensor
        if self.mode == "sparse":
            return self._hook_sparse(params, vis_id, vis_num)
        slot %= self.fb_union.shape[0]
        # 1. union of visibility
        self.mask.zero_()
        self.ops.mark(self.mask[: self.chunks], vis_id, vis_num)
        dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group)
        # 2. ordered union list + inverse map, on the device
        union_ids, union_count, rank = self.ops.compact(self.mask[: self.chunks])
        # 3. size of the packed buffer: predicted from the last visit of this slot, exact (blocking) on the first
        pred = int(self.fb_union[slot])
        if pred <= 0:
            U = max(int(union_count.item()), 1)
        else:
            U = min(self.chunks, int(1.2 * pred) + 1)
        self.ops.feedback(self.fb_union[slot:slot + 1], union_count)
        self.last_alloc = U
        buf = self.flat[: self.nrows * U * self.S].view(self.nrows, U, self.S)
        buf.zero_()
        loc = rank[vis_id.clamp(0, self.chunks - 1)]          # position of each local chunk in the union (entries >= vis_num unused)
        row = 0
        for p, r in zip(params, self.rows):
            g = p.grad
            view = buf[row:row + r]
            if g is not None:
                if hasattr(g, "compacted_values"):
                    self.ops.scatter_add(view, g.compacted_values.reshape(r, -1, self.S), loc, vis_num)
                else:                                           # dense gradient: gather the union rows
                    view.add_(g.reshape(r, self.chunks, self.S)[:, union_ids[:U], :])
            row += r
        if self.use_avg:
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.mul_(1.0 / self.world)
        # 4. compact .grad views over the union
        ids = union_ids[:U]
        row = 0
        for p, r in zip(params, self.rows):
            p.grad = CompactedTensor(p.shape, ids, buf[row:row + r].reshape(*p.shape[:-2], U, self.S))
            row += r
        return ids, union_count


    # -- sparse mode ----------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _hook_sparse(self, params: List[torch.Tensor], vis_id: torch.Tensor, vis_num: torch.Tensor):
        S, chunks, W = self.S, self.chunks, self.world
        A = vis_id.shape[0]
        dev = vis_id.device
        # a. Gaussians of this rank with a non-zero gradient (slots at or beyond vis_num are dirty by design: masked out)
        nz = torch.zeros((A, S), dtype=torch.bool, device=dev)
        comp = []
        for p, r in zip(params, self.rows):
            g = p.grad
            if g is None:
                comp.append(None)
                continue
            v = g.compacted_values.reshape(r, -1, S) if hasattr(g, "compacted_values") else g.reshape(r, chunks, S)[:, vis_id, :]
            comp.append(v)
            nz |= v.any(dim=0)                      # non-zero test without a boolean temporary (NaN counts as non-zero, -0.0 does not)
        nz &= (torch.arange(A, device=dev) < vis_num.to(dev)).unsqueeze(1)
        idx = nz.reshape(-1).nonzero().squeeze(1)                       # host sync: K is needed to size the collective
        K = int(idx.shape[0])
        # b. union of visible chunks and the job's largest K in one small collective
        self.mask.zero_()
        self.ops.mark(self.mask[:chunks], vis_id, vis_num)
        self.mask[chunks] = K
        dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group)
        union_ids, union_count, _ = self.ops.compact(self.mask[:chunks])
        kmax = max(int(self.mask[chunks].item()), 1)
        self.last_k = (K, kmax)
        # c. fixed-size block [nrows + 1, kmax]: the wire container is INT32 (value rows are float bit patterns, the last row the global
        #    Gaussian indices): an integer collective can only copy, whereas float transport of small integers (subnormal patterns)
        #    would be at the mercy of any flush-to-zero on the way.  padding = index 0, value 0
        block = torch.zeros((self.nrows + 1, kmax), dtype=torch.int32, device=dev)
        values = block[: self.nrows].view(torch.float32)
        block[self.nrows, :K] = (vis_id[idx // S] * S + idx % S).to(torch.int32)
        row = 0
        for v, r in zip(comp, self.rows):
            if v is not None:
                values[row:row + r, :K] = v.reshape(r, -1)[:, idx]
            row += r
        gathered = torch.empty((W * (self.nrows + 1), kmax), dtype=torch.int32, device=dev)         # concatenation along dim 0
        dist.all_gather_into_tensor(gathered, block, group=self.group)
        gathered = gathered.view(W, self.nrows + 1, kmax)
        # d. rank-ordered accumulation into the dense gradient (deterministic: replicas stay bit-identical).  The buffer is all zeros
        #    except for the columns the previous exchange added to: clearing those instead of zeroing 708 MB saves ~0.1 ms per step
        dense = self.flat.view(self.nrows, chunks * S)
        if self._touched is not None:
            dense[:, self._touched] = 0.0
        index = gathered[:, self.nrows].long()                                                      # [W, kmax]
        if int(index.max()) >= chunks * S or int(index.min()) < 0:
            raise RuntimeError("litegs_amd.dp: gathered Gaussian index out of range (corrupted exchange block)")
        for w in range(W):
            dense.index_add_(1, index[w], gathered[w, : self.nrows].view(torch.float32), alpha=1.0 / W)
        self._touched = index.reshape(-1)
        row = 0
        for p, r in zip(params, self.rows):
            p.grad = dense[row:row + r].view(p.shape)
            row += r
        return union_ids, union_count


def frame_for(step: int, rank: int, world: int, n_frames: int, perm=None) -> int:
    """Rank r trains frame perm[(step*world + r) mod n_frames]: disjoint frames within a step, same permutation everywhere."""
    k = (step * world + rank) % n_frames
    return int(perm[k]) if perm is not None else k


# ==================================================================================================================================
# mode "moments": the exchange of the native executor (csrc/dp.hip)
# ==================================================================================================================================
class HipMomentOps:
    """Device primitives of the moment exchange on the HIP kernels (no CPU path)."""

    # `spec` (optional, all three): a Speculation record of the exchange -- the step runs under rank-consistent speculative culling
    @staticmethod
    def compact_moments(pg, vis_ids, vis_num, A, S, cap, block, hot_of=None, hot_counter=None, spec=None):
        """hot_of / hot_counter (device addresses, optional): the frame's gradient replicas (csrc/raster.hip) are folded into the records"""
        from ._lib import check, lib
        check(lib().lg_dp_compact_moments_spec(pg.data_ptr(), vis_ids.data_ptr(), vis_num.data_ptr(), A, S, cap, block.data_ptr(), hot_of, hot_counter,
                                               spec.poison.data_ptr() if spec is not None else None,
                                               torch.cuda.current_stream().cuda_stream), "dp_compact_moments")

    @staticmethod
    def build_slotmap(gathered, W, cap, total, slot, host_max_k_ptr, overflow, spec=None):
        from ._lib import check, lib
        check(lib().lg_dp_build_slotmap_spec(gathered.data_ptr(), W, cap, total, slot.data_ptr(), host_max_k_ptr, overflow.data_ptr(),
                                             spec.poison.data_ptr() if spec is not None else None,
                                             spec.status_addr(spec.step_id) if spec is not None else None, spec.step_id if spec is not None else 0,
                                             torch.cuda.current_stream().cuda_stream), "dp_build_slotmap")

    @staticmethod
    def backward_adam(union_ids, union_count, chunks, S, H, Wimg, views, projs, W, degree, R, gathered, cap, slot, ps, ms, vs, lr6, eps, touched=None,
                      spec=None):
        import ctypes
        from ._lib import check, lib
        va = (ctypes.c_float * (16 * W))(*[float(x) for v in views for x in v])
        pa = (ctypes.c_float * (16 * W))(*[float(x) for v in projs for x in v])
        la = (ctypes.c_float * 6)(*lr6)
        check(lib().lg_dp_backward_adam_spec(union_ids.data_ptr(), union_count.data_ptr(), chunks, S, H, Wimg, va, pa, W, degree, R,
                                             gathered.data_ptr(), cap, slot.data_ptr(), *[p.data_ptr() for p in ps], *[m.data_ptr() for m in ms],
                                             *[v.data_ptr() for v in vs], la, 0.9, 0.999, eps, touched.data_ptr() if touched is not None else None,
                                             spec.poison.data_ptr() if spec is not None else None,
                                             spec.applied_addr if spec is not None else None, spec.step_id if spec is not None else 0,
                                             torch.cuda.current_stream().cuda_stream),
              "dp_backward_adam")


class Speculation:
    """What a step under rank-consistent speculative culling hands to the device primitives: the renderer's sticky poison word, the pinned
    word that receives the number of the last step whose Adam ran, and a ring of per-step status words {step number, flags} (flags: bit r =
    rank r's culled forward failed, bit 8 = a record block overflowed: functions of the gathered headers, the same on every rank)."""
    RING = 64
    OVERFLOW = 1 << 8

    def __init__(self, poison: torch.Tensor, applied_addr, words=None):
        self.poison, self.applied_addr = poison, applied_addr
        self.step_id = 0
        if words is None:                      # device stores land here: pinned words of the library's arena (hostwords.py)
            from .hostwords import HostWords
            self._owner = HostWords(2 * self.RING)
            self.words = self._owner.a
            self._addr = self._owner.addr
        else:                                  # CPU tests: a plain integer array, "addresses" are indices
            self.words = words
            self._addr = lambda i: i
        self.words[:] = 0

    def status_addr(self, step_id: int):
        return self._addr(2 * (step_id % self.RING))

    def status(self, step_id: int) -> int:
        """flags of step `step_id` -- call only after the step's event completed"""
        i = 2 * (step_id % self.RING)
        if int(self.words[i]) != step_id:
            raise RuntimeError(f"litegs_amd.dp: status word of step {step_id} holds step {int(self.words[i])} (ring overrun, or the step never ran)")
        return int(self.words[i + 1])

    def close(self):
        owner = getattr(self, "_owner", None)
        if owner is not None:
            owner.close()
            self._owner = None


class LockstepSpeculation:
    """Host side of rank-consistent speculative culling (see MomentExchange): which steps may still have to be replayed, when a step's
    verdict is read, and the replay itself.  Device agnostic -- the owner supplies

      run(record, force)        enqueue the step `record` = (step number, ...) again; force: unculled on this rank
      on_failed(record, flags)  bookkeeping before the forced replay of a failed step (clear the poison word, widen the frame's margin ...)
      event()                   -> an object with synchronize(), recorded behind the step just enqueued
      sync()                    everything enqueued so far has run

    Every rank calls before_step / after_step / flush at the same points of its step sequence; since the verdicts are the same on every
    rank (they are functions of the gathered headers), so are all decisions taken here, and the collectives inside `run` stay matched."""

    def __init__(self, exchange, depth: int, run, on_failed, event, sync):
        self.ex, self.depth = exchange, max(int(depth), 1)
        self.run, self.on_failed, self.event, self.sync = run, on_failed, event, sync
        self.ring = []                 # records of the steps whose verdict has not been read (or that follow a failed one)
        self.events = []               # (step number, event) in step order
        self.replays = 0

    def before_step(self, record) -> None:
        """record[0] = the step number.  Reads the verdicts that are due, replays if one of them is a failure, then registers the step."""
        while len(self.events) >= self.depth:
            self._verify(*self.events.pop(0))
        self.ring.append(record)

    def after_step(self, record) -> None:
        self.events.append((record[0], self.event()))

    def flush(self) -> None:
        while self.events:
            self._verify(*self.events.pop(0))
        self.ring = []

    def _verify(self, no: int, ev) -> None:
        ev.synchronize()
        flags = self.ex.status(no)
        if flags == 0:
            self.ring = [r for r in self.ring if r[0] > no]
            return
        self._recover(no, flags)

    def _recover(self, first_failed: int, flags: int) -> None:
        """From `first_failed` on no replica changed anything.  Replay in order, each step checked before the next: the failed one
        unculled on every rank and with an exact block capacity (MomentExchange.after_failed_step)."""
        self.sync()                    # the steps enqueued behind the failed one have run (as no-ops), their collectives included
        todo = [r for r in self.ring if r[0] >= first_failed]
        self.ring, self.events = [], []
        i, force = 0, True
        while i < len(todo):
            rec = todo[i]
            if force:
                self.on_failed(rec, flags)
            self.ring = [rec]
            self.run(rec, force)
            self.replays += 1
            self.sync()
            again = self.ex.status(rec[0])
            if again != 0:
                if force:
                    raise RuntimeError(f"litegs_amd.dp: step {rec[0]} failed again when replayed unculled with an exact block capacity (flags {again:#x})")
                force, flags = True, again            # a culled step behind the failed one failed as well: same treatment, from here
                continue
            force = False
            i += 1
        self.ring = []


class MomentExchange:
    """Data-parallel step of the native executor: what travels is the blend backward's moment records (csrc/dp.hip header), the
    parameter gradients are rebuilt from them on every rank inside the fused backward + Adam kernel.

    Per step: one small ``all_reduce(MAX)`` (union of the ranks' visible chunks; the chunks some rank saw are the chunks Adam touches)
    and one ``all_gather`` of fixed-size record blocks.  The block capacity follows the GPU-driven sizing protocol
    (litegs/data.py:236-241): the job's largest record count of a slot's previous visit x1.5; only a slot's first visit blocks on a
    count.  Every rank computes that count from the same gathered headers and its device stores it into a pinned word of ITS host;
    the host reads the word only after the event recorded behind the kernel that wrote it (free when the slot recurs an epoch later,
    a wait when two frame sets share a slot back to back) -- so every rank sizes its collective from the same number, whatever the
    relative progress of the hosts.  A slot index beyond ``n_slots`` raises (no silent aliasing of frame sets).
    A count that outgrows the capacity means records were DROPPED: the device marks it in the slot's second pinned word and the
    host raises at the start of the next step but one (the step is named), ``check()`` at the latest -- never silently.
    No ``nonzero()``, no ``.item()`` in the steady state.

    ``profile = True`` brackets the phases of every step with events (``timing()``): wait for the union-of-visibility collective,
    record compaction, all_gather, slot map, backward + Adam over the union.

    **Rank-consistent speculation** (``enable_speculation``; driven by ``FrameTrainer`` with ``speculative = True``).  On one GPU a culled
    step enqueues no gated repeat: a violated depth bound poisons the Adam launches from that step on and the trainer replays them
    (csrc/fused.hip).  Across ranks the same works only if every replica stops updating at the SAME step and every host starts the replay
    at the SAME point of its enqueue sequence -- otherwise the collectives of a replaying rank pair up with those of a rank that is
    still enqueuing new steps.  Both follow from one rule: *the verdict on a step is a function of the gathered headers*.  A rank
    whose culled forward failed sends a flag in its header; every rank's slot-map kernel reads all W headers, raises its own sticky
    poison word when any flag is set -- or when any record count outgrew the block capacity, which turns the overflow from an error
    into a failed step -- and writes the verdict into that step's status words (``Speculation``).  The hosts never look at the sticky
    word (its value at a given wall-clock moment differs between ranks); they read the status of step n - depth behind that step's event
    while enqueuing step n, so all of them find the first failed step s at the same n, with the same steps s+1 .. n-1 enqueued behind it
    (no-ops on every replica, collectives matched).  The replay runs step s unculled on every rank with an exact (blocking) block
    capacity, then the steps behind it as they were -- checking each before the next.  Because an
    overflow is now survivable the capacity shrinks from 1.5 x to ``spec_cap_factor`` = 1.125 x the slot's previous count once that count
    has settled (growth <= 5 % between its last two visits; early in training it doubles between visits and replays would cost more than
    padding): the all_gather moves a quarter less (the padding was a third of the exchanged bytes).
    """

    def __init__(self, params, world: int, ops=HipMomentOps, union_ops=HipOps, group=None, n_slots: int = 64):
        self.world, self.ops, self.union_ops, self.group = world, ops, union_ops, group
        self.n_slots = max(int(n_slots), 1)
        self.cap_factor, self.cap_margin = 1.5, 64       # block capacity = factor x (largest count of the slot's last visit) + margin
        self.spec_cap_factor = 1.125                     # ... under speculation, where an overflow is a replayed step instead of an error,
        self.spec_settled_growth = 1.05                  #     once the slot's count grew by no more than this factor between its last two visits
        self.profile = False
        self.spec = None                                 # Speculation record while rank-consistent speculative culling is on
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.overflow_replays = 0
        self._overflow_at = []
        self._loose_until = 0
        self.rebind(params)

    supports_speculation = True

    def enable_speculation(self, poison: torch.Tensor, applied_addr, words=None) -> "Speculation":
        """poison: the renderer's sticky device word; applied_addr: address of its pinned 'last applied step' word (fast.FusedRenderer)"""
        if self.spec is None or self.spec.poison is not poison:
            if self.spec is not None:
                self.spec.close()
            self.spec = Speculation(poison, applied_addr, words)
        return self.spec

    def disable_speculation(self) -> None:
        if self.spec is not None:
            self.spec.step_id = 0

    def status(self, step_id: int) -> int:
        return self.spec.status(step_id)

    def after_failed_step(self, slot: int, flags: int) -> None:
        """host side of a replay (every rank, same point): the slot map may hold entries of the steps that ran as no-ops; the slot's next
        visit -- the forced replay of the failed step -- sizes its blocks exactly (blocking count): an unculled frame yields another
        record count than the prediction was made for, and the forced replay must not fail"""
        self.slot.zero_()
        self.in_flight = []
        self._wait_slot(slot)
        self.fb_k[slot, 0] = 0
        if flags & Speculation.OVERFLOW:
            self.overflow_replays += 1
            # a run of overflows (counts that keep jumping although they looked settled): give the slack back for 256 steps rather than live on
            # replays.
            # Decided from events every rank sees at the same step, so the ranks keep sizing their blocks alike
            self._overflow_at = [t for t in self._overflow_at if self.steps - t < 64] + [self.steps]
            if len(self._overflow_at) >= 8:
                self._loose_until = self.steps + 256
                self._overflow_at = []

    def rebind(self, params) -> None:
        p0 = params[0]
        self.chunks, self.S = p0.shape[-2], p0.shape[-1]
        dev = p0.device
        self.cuda = bool(p0.is_cuda)
        if self.cuda:
            torch.cuda.current_stream().synchronize()
        if getattr(self, "in_flight", None):             # records dropped by the last steps before the re-bind are reported, never reset away
            self._raise_if_dropped(self.steps)
            if int(self.overflow.item()) != 0:
                raise RuntimeError("litegs_amd.dp: a record block overflowed its predicted capacity; gradients of that step were truncated")
        self.mask = torch.zeros((self.chunks,), dtype=torch.int32, device=dev)
        self.slot = torch.zeros((self.world, self.chunks * self.S), dtype=torch.int32, device=dev)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=dev)
        # per slot: {largest count of the job, overflow marker}, written by the device (csrc/dp.hip: dp_slotmap_kernel)
        # per-slot record counts start over (a blocking first visit per slot): a dropped record is an error here, never a silent truncation,
        # and density control can double a small cloud -- the renderer's table sizes, whose overflow self-heals, are kept instead (fast.py)
        self.fb_k = torch.zeros((self.n_slots, 2), dtype=torch.int32)
        if self.cuda:                                    # library arena: never unmapped under a launch in flight (hostwords.py)
            from .hostwords import pinned_int32
            self.fb_k, self._fb_k_owner = pinned_int32((self.n_slots, 2))
        self.fb_event = [None] * self.n_slots            # recorded behind the kernel that writes fb_k[slot]
        self.prev_k = [0] * self.n_slots                 # the count each slot's last visit was sized from
        self.in_flight = []                              # (step number, slot) of steps whose overflow word has not been read yet
        self.steps = 0
        self.last_cap = 0
        self.last_factor = self.cap_factor
        self._overflow_at, self._loose_until = [], 0      # (step numbers start over)
        self.bytes_last = 0
        self._mask_work = None
        self._marks = []

    def ensure_slots(self, n_slots: int) -> None:
        """grow the per-slot feedback (call before a loop whose epochs have more steps than the constructor was told)"""
        if n_slots > self.n_slots:
            if self.cuda:
                torch.cuda.current_stream().synchronize()
            if self.cuda:
                from .hostwords import pinned_int32
                grown, owner = pinned_int32((n_slots, 2))
            else:
                grown, owner = torch.zeros((n_slots, 2), dtype=torch.int32), None
            grown[: self.n_slots] = self.fb_k
            self.fb_k, self._fb_k_owner = grown, owner
            self.fb_event += [None] * (n_slots - self.n_slots)
            self.prev_k += [0] * (n_slots - self.n_slots)
            self.n_slots = n_slots

    def begin(self, vis_ids, vis_num) -> None:
        """start of a step, as early as the rank's visibility is known (litegs_amd/fast.py calls it right after the culling is
        enqueued): the union-of-visibility all_reduce is launched asynchronously and completes on the collective's stream while the
        forward, the loss and the blend backward run; step() only waits for it.  Optional: without it step() does the same
        collective synchronously."""
        self.mask.zero_()
        self.union_ops.mark(self.mask, vis_ids, vis_num)
        self._mask_work = dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group, async_op=True)

    def _wait_slot(self, slot: int) -> None:
        ev = self.fb_event[slot]
        if ev is not None:
            ev.synchronize()
            self.fb_event[slot] = None

    def _raise_if_dropped(self, upto_step: int, only_slot: int = -1) -> None:
        """reads the overflow words of the steps numbered <= upto_step (their events have normally long completed), or of the step
        that last used `only_slot` (its word is about to be overwritten)"""
        keep = []
        for (no, slot) in self.in_flight:
            if no > upto_step and slot != only_slot:
                keep.append((no, slot))
                continue
            self._wait_slot(slot)
            k = int(self.fb_k[slot, 1])
            if k > 0:
                self.fb_k[slot, 1] = 0
                self.in_flight = [x for x in self.in_flight if x[0] > no]
                raise RuntimeError(f"litegs_amd.dp: step {no} (slot {slot}): {k} moment records exceeded the predicted block capacity; "
                                   "the gradients of that step were truncated")
        self.in_flight = keep

    def check(self) -> None:
        """raises if any step since the last call dropped records (capacity outgrown); synchronises -- call at epoch boundaries"""
        self._raise_if_dropped(self.steps)
        if int(self.overflow.item()) != 0:
            self.overflow.zero_()
            raise RuntimeError("litegs_amd.dp: a record block overflowed its predicted capacity; gradients of that step were truncated")

    def _mark(self, tag: str) -> None:
        if self.profile and self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((self.steps, tag, ev))

    def timing(self) -> dict:
        """mean milliseconds per phase over the profiled steps (synchronises); clears the marks"""
        if self.cuda:
            torch.cuda.current_stream().synchronize()
        by_step = {}
        for no, tag, ev in self._marks:
            by_step.setdefault(no, []).append((tag, ev))
        acc, n = {}, 0
        for no, marks in by_step.items():
            if len(marks) < 2:
                continue
            n += 1
            for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
                acc[t1] = acc.get(t1, 0.0) + e0.elapsed_time(e1)
        self._marks = []
        return {"steps": n, **{k: round(v / max(n, 1), 4) for k, v in acc.items()}}

    @torch.no_grad()
    def step(self, pending: dict, cams, ps, ms, vs, lr6, eps: float, H: int, Wimg: int, slot: int = 0, touched=None, step_id: int = 0):
        """pending: what the blend backward left (litegs_amd/fast.py: pg, A, S, vis_ids, vis_num, degree, chunks, Rr); cams: the W
        ranks' (view_host16, proj_host16) of this step in rank order; ps / ms / vs: parameters and Adam moments in the order
        xyz, scale, rot, sh_0, sh_rest, opacity; touched (nullable uint8[chunks*S]): the optimizer's "has Adam history" flags -- Gaussians
        without history that no rank sent a record for are skipped (an exact no-op); step_id > 0 (with enable_speculation): the step runs
        under rank-consistent speculative culling and reports into the status words of that number.  -> (union_ids, union_count)"""
        W, S, chunks = self.world, self.S, self.chunks
        spec = self.spec if (self.spec is not None and step_id > 0) else None
        sk = {}
        if spec is not None:
            spec.step_id = int(step_id)
            sk = dict(spec=spec)
        A, vis_ids, vis_num, pg = pending["A"], pending["vis_ids"], pending["vis_num"], pending["pg"]
        if not 0 <= slot < self.n_slots:
            raise ValueError(f"MomentExchange: slot {slot} outside [0, {self.n_slots}) -- construct it with n_slots = steps per epoch")
        self.steps += 1
        if spec is None:
            self._raise_if_dropped(self.steps - 2)       # steps older than the previous one: their words have landed
        dev = pg.device
        hot = {}                                         # gradient replicas of this frame (litegs_amd/fast.py): folded by the compaction
        if pending.get("replicas"):
            hot = dict(hot_of=pending["hot_of"], hot_counter=pending["hot_counter"])
        nrec = self.ops.record_floats() if hasattr(self.ops, "record_floats") else 10
        self._mark("start")
        # union of visibility (device-side list + count): already in flight if begin() was called for this step
        work = getattr(self, "_mask_work", None)
        if work is not None:
            work.wait()
            self._mask_work = None
        else:
            self.mask.zero_()
            self.union_ops.mark(self.mask, vis_ids, vis_num)
            dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group)
        self._mark("union_wait_ms")
        union_ids, union_count, _ = self.union_ops.compact(self.mask)
        # capacity of the record blocks: the count every rank's device derived from the gathered headers of the slot's last visit
        if spec is None:
            self._raise_if_dropped(0, only_slot=slot)
        self._wait_slot(slot)
        pred = int(self.fb_k[slot, 0])
        exact = pred <= 0
        if exact:                                        # first visit of the slot (or the replay of an overflowed step): blocking count
            probe = torch.empty(((1 + A * S) * nrec,), dtype=torch.int32, device=dev)
            self.ops.compact_moments(pg, vis_ids, vis_num, A, S, A * S, probe, **hot)
            k = probe[:1].clone()
            dist.all_reduce(k, op=dist.ReduceOp.MAX, group=self.group)
            pred = max(int(k.item()), 1)
        # how fast the slot's count has been moving (host-side memory of the values read here: the same on every rank)
        last = self.prev_k[slot]
        settled = last > 0 and pred <= self.spec_settled_growth * last
        self.prev_k[slot] = pred
        if exact:
            cap = pred + self.cap_margin                 # the count IS this step's: no slack needed
        else:
            # under speculation an overflow costs a replay, not the run: the slack shrinks once the slot's count has stopped moving
            # (early in training it doubles between visits: there the replays would cost more than the padding)
            tight = spec is not None and settled and self.steps >= self._loose_until
            cap = int((self.spec_cap_factor if tight else self.cap_factor) * pred) + self.cap_margin
            self.last_factor = self.spec_cap_factor if tight else self.cap_factor
        self.last_cap = cap
        # wire container: int32 words (record = index word + nine float bit patterns) -- an integer collective can only copy
        block = torch.empty(((1 + cap) * nrec,), dtype=torch.int32, device=dev)
        self.ops.compact_moments(pg, vis_ids, vis_num, A, S, cap, block, **hot, **sk)
        self._mark("compact_ms")
        gathered = torch.empty((W * (1 + cap) * nrec,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered, block, group=self.group)
        self._mark("all_gather_ms")
        self.bytes_last = (W - 1) * block.numel() * 4
        self.ops.build_slotmap(gathered, W, cap, chunks * S, self.slot, self.fb_k.data_ptr() + 8 * slot, self.overflow, **sk)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.fb_event[slot] = ev
        if spec is None:
            self.in_flight.append((self.steps, slot))
        self._mark("slotmap_ms")
        self.ops.backward_adam(union_ids, union_count, chunks, S, H, Wimg, [c[0] for c in cams], [c[1] for c in cams], W, pending["degree"],
                               pending["Rr"], gathered, cap, self.slot, ps, ms, vs, lr6, eps, touched, **sk)
        self._mark("backward_adam_ms")
        return union_ids, union_count
