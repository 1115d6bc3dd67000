"""One training iteration of the LiteGS hot loop (litegs/training/trainer.py:111-163):
render_preprocess -> render -> L1+SSIM loss -> backward -> sparse Adam -> zero_grad -> lr schedule.

``FrameTrainer`` owns the parameters, a set of camera frames with per-frame targets and the pinned feedback buffers of the
GPU-driven protocol (litegs/data.py:236-241), so steady-state iterations run without any host<->device synchronisation.
``SyntheticTrainer`` (bench.py, the smoke test, the DP tests) fills it with a seeded cloud; ``litegs_amd.training.start`` with a COLMAP
scene.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import fast
from . import loss as loss_mod
from . import optimizer as opt_mod
from . import render as R
from . import synthetic as S
from .statistics import STATS


class Frame:
    def __init__(self, view, proj, planes, gt, idx):
        self.view, self.proj, self.planes, self.gt = view, proj, planes, gt
        self.idx_tensor = torch.tensor([idx], dtype=torch.int64)       # CPU, as the reference's DataLoader yields it
        self.cam = fast.CameraFrame(view, proj, planes, idx)


class FrameTrainer:
    """The hot loop over a given set of posed frames and a given parameter set (the reference's per-iteration body,
    trainer.py:121-163).  ``SyntheticTrainer`` (seeded cloud + orbit cameras, bench / tests) and ``litegs_amd.training.start`` (COLMAP
    scenes) are its two callers."""

    def __init__(self, params: List[torch.nn.Parameter], frames: List[Frame], height: int, width: int, opt, sched, pp=None,
                 sh_degree: int = 3, device: Optional[torch.device] = None, loss_fn=None, fused: bool = True,
                 fuse_adam: bool = True, extra_slots: int = 0):
        """fused=True: native executor (litegs_amd/fast.py); fused=False: operator-by-operator path through the litegs_fused surface.
        loss_fn: None = the HIP L1+SSIM loss (csrc/loss.hip); a callable (img[1,3,H,W] in [0,1], gt) -> scalar replaces it (tests).
        extra_slots: additional per-frame feedback slots behind the training frames' (evaluation frames, indices len(frames)...)."""
        self.device = device or params[0].device
        self.H, self.W, self.degree = height, width, sh_degree
        self.pp = pp or R.PipelineParams()
        self.params = list(params)
        self.n_chunks, self.S = self.params[0].shape[-2], self.params[0].shape[-1]
        self.frames = frames
        n_frames = len(frames) + extra_slots
        # Implicit synchronisation buffers of the operator path: written in epoch N, read in epoch N+1 (litegs/data.py:238).  Pinned words
        # from the library's arena (hostwords.py): the device stores into them asynchronously, so they must never be unmapped under it
        from .hostwords import HostWords
        self._fb_words = HostWords(2 * n_frames)
        self.feedback_visible_chunks_num = torch.from_numpy(self._fb_words.a[:n_frames])
        self.feedback_binning_allocate_size = torch.from_numpy(self._fb_words.a[n_frames:])
        self.opt, self.sched = opt, sched
        with torch.no_grad():
            xyz, scale, rot = self.params[0], self.params[1], self.params[2]
            self.cluster_origin, self.cluster_extend = R.get_cluster_AABB(xyz, scale.exp(), torch.nn.functional.normalize(rot, dim=0))
        self.loss_fn = loss_fn if loss_fn is not None else loss_mod.fused_l1_ssim_loss
        self.fused = fused
        # the native executor hands the raw raster image to the loss kernels (crop + clamp fused in); only with the HIP loss
        self.raw_loss = fused and loss_fn is None
        self._unit = torch.ones((), dtype=torch.float32, device=self.device)      # d(loss)/d(loss): reused, no fill launch per step
        self.renderer = fast.FusedRenderer(n_frames, height, width, self.pp.tile_size, self.pp.cluster_size)
        self.fadam = fast.FusedAdam(self.opt, self.renderer)
        self.fuse_adam = fuse_adam
        # lr-schedule ticks per step: a data-parallel step consumes `world` frames of the reference's iteration budget
        # (litegs/training/trainer.py:108: total_epoch = iterations / frames), so training.start sets this to the world size and the
        # position learning rate reaches its final value after the same number of FRAMES as on one GPU
        self.sched_ticks = 1
        self.last = {}
        # speculative culling (csrc/fused.hip): culled steps enqueue no gated repeat; a failed step poisons the Adam launches from there on
        # and is replayed here, unculled, together with the steps enqueued behind it.  Parameters are final only after flush().
        self.speculative = False
        self._spec_ring = []          # (step number, frame index, learning rates) of the steps that may still have to be replayed
        self._spec_next = 1
        self._spec_events = []        # one event behind every speculative step still in flight
        self._spec_exchange = None    # the dp.MomentExchange the speculative steps in flight went through (None: single GPU)
        self._spec_dp = None          # ... and their bookkeeping (dp.LockstepSpeculation)
        # Across ranks a failed forward costs EVERY rank a synchronisation and the replay of up to three steps, the gated repeat 40 us per
        # step: with more than one failure per 64 steps the forwards go back to the gated repeat for a while (the exchange stays
        # speculative: an overflowing record block is still a replayed step).  Decided from the job-wide verdicts: the same on every rank.
        self.dp_fail_window, self.dp_fail_limit, self.dp_gated_steps = 64, 2, 256
        self._dp_fail_steps = []
        self._dp_gated_until = 0
        self.dp_gated_periods = 0
        self.spec_depth = 2           # steps the host may run ahead of the device in speculative mode
        self.spec_replays = 0
        self.spec_log = []                # (step number, frame, visits of that frame so far, steps replayed) per violated bound
        self.sanitised = {}               # site -> garbage table words neutralised by a kernel so far (csrc/lg_sanity.h); empty = none

    # -------------------------------------------------------------------------------------------
    def forward(self, frame: Frame, raw: bool = False):
        xyz, scale, rot, sh_0, sh_rest, opacity = self.params
        STATS.current_frame = int(frame.idx_tensor[0])
        if self.fused:
            fn = self.renderer.render_raw if raw else self.renderer.render
            img, vis_id, vis_num = fn(frame.cam, self.cluster_origin, self.cluster_extend, xyz, scale, rot, sh_0, sh_rest, opacity, self.degree)
            return img, vis_id, vis_num, None
        vis_id, vis_num, cx, cs, cr, cc, co = R.render_preprocess(
            self.cluster_origin, self.cluster_extend, frame.planes, frame.view, xyz, scale, rot, sh_0, sh_rest, opacity,
            self.feedback_visible_chunks_num, frame.idx_tensor, self.pp, self.degree)
        valid_length = vis_num * self.pp.cluster_size
        img, trans, depth, normal, prim_vis = R.render(frame.view, frame.proj, cx, cs, cr, cc, co, valid_length,
                                                       self.feedback_binning_allocate_size, frame.idx_tensor, self.degree,
                                                       (self.H, self.W), self.pp)
        return img, vis_id, vis_num, prim_vis

    def step(self, frame_index: int, grad_hook=None, hook_slot: int = 0, peer_frames=None):
        """grad_hook: None, a gradient hook (dp.GradientExchange.hook: parameter gradients are materialised and exchanged) or a
        dp.MomentExchange (native executor only: the blend backward's moment records are exchanged, backward + Adam stay fused;
        peer_frames = the frame indices of ranks 0..W-1 of this step)."""
        moments = grad_hook is not None and hasattr(grad_hook, "step") and not callable(grad_hook)
        dp_spec = moments and getattr(grad_hook, "supports_speculation", False)
        spec = (self.speculative and (grad_hook is None or dp_spec) and self.fused and self.fuse_adam and self.raw_loss and not STATS.active)
        if self._spec_pending and (not spec or (self._spec_exchange is not (grad_hook if dp_spec else None))):
            self.flush()                                  # leaving the speculative regime (or changing its kind): everything enqueued must have landed
        if spec:
            self.renderer.enable_speculation(self.device)
            if dp_spec:
                # across ranks: the verdict on step n - depth is read behind that step's event, at the same point of every rank's enqueue
                # sequence (litegs_amd/dp.py "rank-consistent speculation"); never the sticky word, whose value at a wall-clock moment differs
                if self._spec_exchange is not grad_hook:
                    self._spec_dp_start(grad_hook)
                R = self.renderer
                if R.spec_forward:
                    recent = [t for t in self._dp_fail_steps if self._spec_next - t < self.dp_fail_window]
                    if len(recent) >= self.dp_fail_limit:
                        R.spec_forward = False
                        self._dp_gated_until = self._spec_next + self.dp_gated_steps
                        self.dp_gated_periods += 1
                elif self._spec_next >= self._dp_gated_until:
                    R.spec_forward = True
                    self._dp_fail_steps = []
                rec = (self._spec_next, frame_index, [float(g["lr"]) for g in self.opt.param_groups], hook_slot, peer_frames)
                self._spec_dp.before_step(rec)
                self._spec_dp_restore_lrs()
            else:
                self.renderer.spec_forward = True
                # The host must not run far ahead of the device: everything enqueued behind a failed step is wasted and replayed.  Two steps
                # in flight keep the device busy (enqueueing a step takes a third of its run time); the wait is on the step before those.
                if len(self._spec_events) >= self.spec_depth:
                    self._spec_events.pop(0).synchronize()
                self._spec_poll()
            self.renderer.spec_step = self._spec_next
            if not dp_spec:
                self._spec_ring.append((self._spec_next, frame_index, [float(g["lr"]) for g in self.opt.param_groups], hook_slot, peer_frames))
            self._spec_next += 1
            if len(self._spec_ring) > 64 and not dp_spec:  # steps whose Adam launch has reported in need no replay any more
                done = self.renderer.applied_step()
                self._spec_ring = [r for r in self._spec_ring if r[0] > done]
        else:
            self.renderer.disable_speculation()
            self.renderer.spec_forward = True
            if moments and hasattr(grad_hook, "disable_speculation"):
                grad_hook.disable_speculation()
        loss = self._step_body(frame_index, grad_hook, hook_slot, peer_frames)
        if spec and dp_spec:
            self._spec_dp.after_step(rec)
        elif spec:
            ev = torch.cuda.Event()
            ev.record()
            self._spec_events.append(ev)
        for _ in range(self.sched_ticks):
            self.sched.step()
        return loss

    @property
    def _spec_pending(self) -> bool:
        """speculative steps are in flight whose verdict has not been read: the parameters are final only after flush()"""
        return bool(self._spec_ring) or self._spec_exchange is not None

    # -- speculative culling across ranks (dp.MomentExchange): verdicts at a fixed lag, replay in lock-step (dp.LockstepSpeculation) ---
    def _spec_dp_start(self, exchange):
        from . import dp
        R = self.renderer
        exchange.enable_speculation(R.spec_poison, R.spec_words.addr(1))
        state = {}

        def on_failed(rec, flags):
            no, frame_index, _lrs, slot, _peers = rec
            R.clear_poison()
            exchange.after_failed_step(slot, flags)
            if flags & 0xff:                              # a culled forward failed on some rank (not a mere block overflow)
                self._dp_fail_steps.append(int(no))
            if (flags >> exchange.rank) & 1:              # this rank's bounds were violated: what the gated repeat's bookkeeping does
                k = self.frames[frame_index % len(self.frames)].cam.index
                if len(self.spec_log) < 64:
                    self.spec_log.append((int(no), int(k), int(R.frames[k].visits), 1))
                R.note_fallback(k)

        def run(rec, force):
            no, frame_index, lrs, slot, peers = rec
            if "lrs" not in state:                        # the learning rates of the step being enqueued come back after the replay
                state["lrs"] = [float(g["lr"]) for g in self.opt.param_groups]
            for g, lr in zip(self.opt.param_groups, lrs):
                g["lr"] = lr
            R.spec_step = no
            R.force_full = force
            self._step_body(frame_index, exchange, slot, peers)
            self.spec_replays += 1

        def event():
            ev = torch.cuda.Event()
            ev.record()
            return ev

        def sync():
            torch.cuda.current_stream().synchronize()

        self._spec_dp = dp.LockstepSpeculation(exchange, self.spec_depth, run, on_failed, event, sync)
        self._spec_dp_state = state
        self._spec_exchange = exchange

    def _spec_dp_restore_lrs(self):
        lrs = self._spec_dp_state.pop("lrs", None)
        if lrs is not None:
            for g, lr in zip(self.opt.param_groups, lrs):
                g["lr"] = lr

    # -- speculative culling: notice, replay ------------------------------------------------------------------------------------
    def _spec_poll(self):
        if self.renderer.poisoned():
            self._spec_recover()

    def _spec_recover(self):
        """a culled step failed: from that step on no Adam launch changed anything.  Replay them in order -- the first one unculled --
        with the learning rates they were enqueued with."""
        R = self.renderer
        while True:
            torch.cuda.current_stream().synchronize()
            if not R.poisoned():
                break
            done = R.applied_step()
            todo = [r for r in self._spec_ring if r[0] > done]
            self._spec_ring = []
            self._spec_events = []
            R.clear_poison()
            torch.cuda.current_stream().synchronize()
            current = [float(g["lr"]) for g in self.opt.param_groups]
            for i, (no, frame_index, lrs, _slot, _peers) in enumerate(todo):
                for g, lr in zip(self.opt.param_groups, lrs):
                    g["lr"] = lr
                R.spec_step = no
                R.force_full = (i == 0)
                if i == 0:                                # the frame whose bounds were violated: what the gated repeat's bookkeeping does
                    k = self.frames[frame_index % len(self.frames)].cam.index
                    if len(self.spec_log) < 64:
                        self.spec_log.append((int(no), int(k), int(R.frames[k].visits), len(todo)))
                    R.note_fallback(k)
                self._spec_ring.append((no, frame_index, lrs, 0, None))
                self._step_body(frame_index, None, 0, None)
                self.spec_replays += 1
            for g, lr in zip(self.opt.param_groups, current):
                g["lr"] = lr

    def flush(self):
        """the parameters reflect every step enqueued so far (speculative mode: failed steps are replayed first); synchronises"""
        torch.cuda.current_stream().synchronize()
        if self._spec_exchange is not None:               # across ranks: the verdicts not read yet, in step order (every rank flushes at the same step)
            self._spec_dp.flush()
            self._spec_dp_restore_lrs()
            self._spec_exchange = None
        elif self.renderer.spec_words is not None:
            self._spec_recover()
        self._spec_ring = []
        self._spec_events = []
        self.renderer.check_tables()
        self._collect_sanitised()

    def _collect_sanitised(self):
        """lg_sanity.h: a kernel neutralised a garbage table word since the last flush -> counted per site, reported once per site on stderr
        (the run stays alive; `sanitised` is reported by bench.py and the convergence scripts)"""
        if not self.fused:
            return
        counts = self.renderer.sanitised_counts(reset=True)
        for k, v in counts.items():
            if v:
                first = self.sanitised.get(k, 0) == 0
                self.sanitised[k] = self.sanitised.get(k, 0) + v
                if first and k != "truncated_tables":
                    import sys
                    print(f"[litegs_amd] WARNING: {v} table word(s) neutralised at site '{k}' since the last flush -- a table held garbage "
                          f"(DESIGN.md section 9); set LITEGS_VALIDATE_TABLES=1 to locate it", file=sys.stderr, flush=True)

    def _step_body(self, frame_index: int, grad_hook=None, hook_slot: int = 0, peer_frames=None):
        frame = self.frames[frame_index % len(self.frames)]
        moments = grad_hook is not None and hasattr(grad_hook, "step") and not callable(grad_hook)
        if moments and not (self.fused and self.fuse_adam):
            raise RuntimeError("MomentExchange needs the native executor (fused=True)")
        # gradients are only materialised when something consumes them between backward and the optimizer (gradient-hook DP exchange)
        self.renderer.fuse_optimizer = self.fused and self.fuse_adam and (grad_hook is None or moments)
        self.renderer.after_cull = grad_hook.begin if (moments and hasattr(grad_hook, "begin")) else None
        # gradient replicas whenever every consumer of the records folds them: the fused backward + Adam and the moment exchange's
        # compaction (csrc/dp.hip) do
        self.renderer.fold_only_consumer = grad_hook is None or moments
        img, vis_id, vis_num, prim_vis = self.forward(frame, raw=self.raw_loss)
        if self.raw_loss:
            from . import loss_hip
            loss = loss_hip.raster_l1_ssim_loss(img, frame.gt, value_in_backward=True)      # backward() follows at once
        else:
            loss = self.loss_fn(img, frame.gt)
        loss.backward(self._unit)
        if moments:                          # data-parallel moment exchange + fused backward/Adam over the union (csrc/dp.hip)
            peers = peer_frames if peer_frames is not None else [frame_index]
            cams = [(self.frames[f % len(self.frames)].cam.view_host, self.frames[f % len(self.frames)].cam.proj_host) for f in peers]
            vis_id, vis_num = self.fadam.step_exchange(grad_hook, cams, hook_slot)
        elif grad_hook is not None:          # data-parallel gradient exchange (litegs_amd/dp.py)
            vis_id, vis_num = grad_hook(self.params, vis_id, vis_num, hook_slot)
        if moments:
            pass
        elif self.fused:
            self.fadam.step(vis_id, vis_num)
        else:
            self.opt.step(vis_id, vis_num, prim_vis)
        self.opt.zero_grad(set_to_none=True)
        self.last = dict(loss=loss.detach(), vis_num=vis_num)
        return loss

    def close(self):
        """everything enqueued has landed (failed speculative steps replayed) and the renderer's words are released; idempotent"""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        try:
            if torch.cuda.is_available():
                self.flush()
        finally:
            self.renderer.close()
            self._fb_words.close()

    def __del__(self):
        # no replay of failed speculative steps from a finaliser (that is close()'s / flush()'s job): only make sure nothing is in flight
        try:
            if not getattr(self, "_closed", False):
                self._closed = True
                self.renderer.close(sync=False)           # no device sync from a garbage collection: the arena quarantines the words
                self._fb_words.close()
        except Exception:
            pass

    @torch.no_grad()
    def forward_only(self, frame_index: int):
        if self._spec_pending:
            self.flush()
        frame = self.frames[frame_index % len(self.frames)]
        return self.forward(frame)[0]

    # -- epoch-boundary callers of the path (trainer.py:111-118, 195): Morton re-sort, density control ---------------------------
    def _rebind(self):
        """parameters were replaced / re-sorted: refresh everything derived from them (chunk AABBs, cached pointers, depth bounds and tile
        schedules).  The per-frame SIZE predictions of the GPU-driven protocol are kept, as in the reference."""
        if self._spec_pending:
            self.flush()
        by_name = {g["name"]: g["params"][0] for g in self.opt.param_groups}
        self.params = [by_name[n] for n in ("xyz", "scale", "rot", "sh_0", "sh_rest", "opacity")]
        old_chunks = self.n_chunks
        self.n_chunks, self.S = self.params[0].shape[-2], self.params[0].shape[-1]
        with torch.no_grad():
            xyz, scale, rot = self.params[0], self.params[1], self.params[2]
            self.cluster_origin, self.cluster_extend = R.get_cluster_AABB(xyz, scale.exp(), torch.nn.functional.normalize(rot, dim=0))
        self.fadam._ready = False
        self.renderer.pending = None
        torch.cuda.current_stream().synchronize()               # pinned feedback words may still be in flight
        # bounds and schedules dropped; size predictions of the frames in use kept and scaled by the cloud's growth (as the reference does)
        self.renderer.parameters_replaced(self.n_chunks / max(old_chunks, 1))
        if getattr(self, "exchange", None) is not None:
            self.exchange.rebind(self.params)

    def enable_densify(self, params=None, total_epochs: int = 100, screen_extent: float = 1.0, seed: int = 0, group=None,
                       init_points_num: Optional[int] = None):
        from . import densify as D
        dp = params or D.DensifyParams()
        dp.resolve_until(total_epochs)
        self.controller = D.DensityController(screen_extent, dp, self.S, init_points_num or self.n_chunks * self.S, STATS, D.Sampler(seed), group)
        self.controller.on_change = self._rebind
        STATS.reset(self.n_chunks, self.S, self.controller.is_densify_actived, device=self.device)
        return self.controller

    def begin_epoch(self, epoch: int):
        """Morton re-sort one epoch after every densification (trainer.py:113-116); returns the statistics guard for the epoch."""
        if self._spec_pending:
            self.flush()
        ctl = getattr(self, "controller", None)
        if ctl is not None and (epoch - 1) % ctl.p.densification_interval == 0:
            from . import scene
            scene.spatial_refine(True, self.opt, self.params[0])
            self._rebind()
        return STATS.epoch(epoch)

    def end_epoch(self, epoch: int):
        if self._spec_pending:
            self.flush()
        ctl = getattr(self, "controller", None)
        if ctl is not None:
            ctl.step(self.opt, epoch)

    def workload_stats(self, frame_index: int = 0):
        """N_vis (Gaussians after chunk culling), I (tile instances) for one frame -- host sync, call outside timed regions."""
        frame = self.frames[frame_index % len(self.frames)]
        if self._spec_pending:
            self.flush()
        torch.cuda.synchronize()
        k = int(frame.idx_tensor[0])
        if self.fused:
            return dict(n_vis=int(self.renderer.fb_vis[k]) * self.S, instances=int(self.renderer.fb_total[k]))
        return dict(n_vis=int(self.feedback_visible_chunks_num[k]) * self.S, instances=int(self.feedback_binning_allocate_size[k]))


class SyntheticTrainer(FrameTrainer):
    """A seeded Gaussian cloud (SURVEY.md 8d distribution), orbit cameras and per-frame noise targets: the bench / test workload."""

    def __init__(self, n_gaussians: int, width: int, height: int, focal: float, n_frames: int = 8, seed: int = 0, sh_degree: int = 3,
                 device: Optional[torch.device] = None, radius: float = 4.0, cam_radius_frac: float = 0.5, loss_fn=None,
                 scene=None, fused: bool = True, fuse_adam: bool = True, noise_targets: bool = True):
        device = device or torch.device("cuda", torch.cuda.current_device())
        if scene is None:
            scene = S.make_scene(n_gaussians, seed=seed, sh_degree=sh_degree, radius=radius)
        params = [torch.nn.Parameter(torch.from_numpy(p).to(device)) for p in scene]
        cams = S.orbit_cameras(n_frames, width, height, focal, focal, cam_radius_frac * radius)
        rng = np.random.default_rng(seed + 1)
        frames: List[Frame] = []
        for k, (view, proj, planes) in enumerate(cams):
            # noise_targets=False: the caller assigns frames[k].gt itself (teacher renders, tests/convergence*.py)
            gt = torch.from_numpy(rng.random((1, 3, height, width), dtype=np.float32)).to(device) if noise_targets else None
            frames.append(Frame(*[torch.from_numpy(x).to(device) for x in (view, proj, planes)], gt, k))
        opt, sched = opt_mod.get_optimizer(*params, 1.0, opt_mod.OptimizationParams())
        super().__init__(params, frames, height, width, opt, sched, None, sh_degree, device, loss_fn, fused, fuse_adam)


def train(trainer: FrameTrainer, epochs: int, exchange=None, rank: int = 0, world: int = 1, start_epoch: int = 0, on_epoch=None):
    """The reference's epoch loop (trainer.py:108-195) around the hot path, data-parallel when ``exchange`` (dp.MomentExchange or
    dp.GradientExchange) is given: each step trains ``world`` different frames (one per rank), gradients are averaged over the union
    of the ranks' visible chunks, and at the epoch boundaries every rank performs the same Morton re-sort and the same density-control
    decisions (statistics summed across ranks, shared random draws) -- replicas stay bit-identical without ever broadcasting
    parameters.  The exchange is re-bound only when the parameters were replaced (density control, re-sort), not every epoch."""
    from . import dp
    n_frames = len(trainer.frames)
    steps_per_epoch = (n_frames + world - 1) // world
    step = start_epoch * steps_per_epoch
    trainer.exchange = exchange
    moments = exchange is not None and not hasattr(exchange, "hook")
    if moments and hasattr(exchange, "ensure_slots"):
        exchange.ensure_slots(steps_per_epoch)           # one feedback slot per frame set of an epoch, never shared (litegs_amd/dp.py)
    for epoch in range(start_epoch, epochs):
        with trainer.begin_epoch(epoch):
            for k in range(steps_per_epoch):
                peers = [dp.frame_for(step, r, world, n_frames) for r in range(world)]
                hook = None if exchange is None else (exchange if moments else exchange.hook)
                trainer.step(peers[rank], hook, k, peers)
                step += 1
        if moments:
            exchange.check()                             # before end_epoch: a re-bind (density control) resets the exchange's overflow words
        trainer.end_epoch(epoch)
        if on_epoch is not None:
            on_epoch(epoch, trainer)
    return trainer
