"""Data-parallel exchange on the real device primitives (HipOps) with a single-rank RCCL group: with one rank the exchange must be
an identity on the gradients (packed over the union = the rank's own visible list), so a hooked trainer has to track an un-hooked one.  (The N>1 logic is covered on CPU by
tests/test_dp_gloo.py; multi-GPU runs are the driver's.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_exchange_is_identity():
    import torch.distributed as dist
    from litegs_amd import dp, synthetic as S
    from litegs_amd.trainer import SyntheticTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = S.make_scene(6000, seed=4)
        ta = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        tb = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        tc = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        hook = dp.GradientExchange(tb.params, 1, mode="dense").hook
        hook_sparse = dp.GradientExchange(tc.params, 1).hook              # default: sparse all_gather exchange
        for i in range(4):
            la = ta.step(i)
            lb = tb.step(i, hook, i % 2)
            lc = tc.step(i, hook_sparse, i % 2)
            assert abs(la.item() - lb.item()) < 1e-5 and abs(la.item() - lc.item()) < 1e-5
        moved = 0.0
        for pa, pb, pc, p0 in zip(ta.params, tb.params, tc.params, scene):
            # independent runs (float atomics reorder the sums; a near-zero gradient's sign decides a ~3 lr Adam step): all but a few
            # elements agree closely, none is far off
            for other in (pb, pc):
                d = (pa - other).abs()
                assert (d > 5e-4).float().mean().item() < 1e-3 and d.max().item() < 0.2, (d.max().item(), (d > 5e-4).float().mean().item())
            moved = max(moved, (pa.detach().cpu() - torch.from_numpy(p0)).abs().max().item())
        assert moved > 1e-4, "parameters must actually have been updated"
        # union list == own visible set, ascending
        fr = tb.frames[0]
        img, vis_id, vis_num, _ = tb.forward(fr)
        img.sum().backward()
        n = int(vis_num.item())
        local = [p.grad.compacted_values.clone() for p in tb.params]
        uid, ucnt = hook(tb.params, vis_id, vis_num, 0)
        assert int(ucnt.item()) == n
        assert torch.equal(uid[:n], vis_id[:n])
        # one rank: the packed union buffer must hold exactly this rank's compact gradients
        for p, g0 in zip(tb.params, local):
            assert p.grad.shape == p.shape and hasattr(p.grad, "compacted_values")
            rows = g0.numel() // (g0.shape[-2] * g0.shape[-1])
            a = p.grad.compacted_values.reshape(rows, -1, g0.shape[-1])[:, :n]
            assert torch.equal(a, g0.reshape(rows, -1, g0.shape[-1])[:, :n])
    finally:
        dist.destroy_process_group()


def test_single_rank_moment_exchange_equals_the_fused_step():
    """world 1: the moment exchange (compaction -> gather of one block -> slot map -> fused backward + Adam over the union) must
    reproduce the single-GPU fused step -- same chain arithmetic, same Adam, same chunks; the two runs differ only by the order of the
    blend backward's float atomics (last-bit differences in the moments)"""
    import torch.distributed as dist
    from litegs_amd import dp, synthetic as S
    from litegs_amd.trainer import SyntheticTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scene = S.make_scene(6000, seed=4)
        ta = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        tb = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)
        ex = dp.MomentExchange(tb.params, 1)
        tc = SyntheticTrainer(6000, 320, 200, 300.0, n_frames=2, scene=scene)       # the same over RCCL under rank-consistent speculation
        tc.speculative = True
        exc = dp.MomentExchange(tc.params, 1)
        for i in range(6):
            la = ta.step(i)
            lb = tb.step(i, ex, i % 2, [i])
            lc = tc.step(i, exc, i % 2, [i])
            assert abs(la.item() - lb.item()) < 1e-6 and abs(la.item() - lc.item()) < 1e-6
        ex.check()
        tc.flush()
        exc.check()
        assert exc.spec is not None and tc.renderer.applied_step() == 6 and int(tc.renderer.spec_poison.item()) == 0
        assert ex.last_cap > 0 and int(ex.slot.abs().sum().item()) == 0, "the slot map must be left clean"
        assert int(exc.slot.abs().sum().item()) == 0
        for pb, pc in zip(tb.params, tc.params):
            d = (pb - pc).abs()
            assert (d > 2e-5).float().mean().item() < 1e-3 and d.max().item() < 0.1
        for pa, pb in zip(ta.params, tb.params):
            # two independent runs: the blend backward's float atomics reorder the sums, and Adam without bias correction turns a
            # near-zero gradient of either sign into a step of ~3 lr -- a handful of elements may differ by that much (and their
            # moments with them), everything else agrees
            d = (pa - pb).abs()
            assert (d > 2e-5).float().mean().item() < 1e-3 and d.max().item() < 0.1, (d.max().item(), (d > 2e-5).float().mean().item())
            ma, mb = ta.opt.state[pa], tb.opt.state[pb]
            for key, rtol in (("exp_avg", 1e-4), ("exp_avg_sq", 1e-3)):
                a, b = ma[key], mb[key]
                bad = (a - b).abs() > rtol * b.abs() + 1e-5 * float(a.abs().max()) + 1e-20
                assert bad.float().mean().item() < 1e-3, (key, bad.float().mean().item())
    finally:
        dist.destroy_process_group()


def _two_rank_moments_worker(rank, world, port, out):
    """two ranks on cuda:0 over gloo: the moment exchange against the gradient exchange (dense mode) on identical replicas"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from litegs_amd import dp, synthetic as S
        from litegs_amd.trainer import SyntheticTrainer
        scene = S.make_scene(20000, seed=4)
        ta = SyntheticTrainer(20000, 320, 200, 700.0, n_frames=2 * world, scene=scene)
        tb = SyntheticTrainer(20000, 320, 200, 700.0, n_frames=2 * world, scene=scene)
        ex_m = dp.MomentExchange(ta.params, world)
        ex_g = dp.GradientExchange(tb.params, world, mode="dense")
        checks = {}
        nframes = len(ta.frames)
        for i in range(4):
            peers = [dp.frame_for(i, r, world, nframes) for r in range(world)]
            ta.step(peers[rank], ex_m, i % 2, peers)
            tb.step(peers[rank], ex_g.hook, i % 2)
        ex_m.check()
        torch.cuda.synchronize()
        checks["slot_clean"] = int(ex_m.slot.abs().sum().item()) == 0
        checks["bytes"] = ex_m.bytes_last > 0
        worst = 0.0
        for pa, pb, nm in zip(ta.params, tb.params, ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]):
            d = (pa - pb).abs()
            worst = max(worst, float((d > 5e-4).float().mean()))
            checks["none_far_off_" + nm] = float(d.max()) < 0.2
        checks["matches_gradient_exchange"] = worst < 1e-3          # fraction of elements beyond 5e-4 (see test_single_rank_exchange_is_identity)
        checks["worst"] = worst
        flat = torch.cat([p.detach().reshape(-1) for p in ta.params]).cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        checks["replicas_identical"] = torch.equal(both[0], both[1])
        checks["moved"] = not torch.equal(flat, torch.cat([torch.from_numpy(a).reshape(-1) for a in scene]))
        out[rank] = checks
    finally:
        dist.destroy_process_group()


def test_two_ranks_moment_exchange_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_moments_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        c = dict(out.get(r) or {})
        assert c and all(v for k, v in c.items() if k != "worst"), (r, c)


def _dense_grad(p, n_valid, chunks, S):
    g = p.grad
    return (g.to_dense(n_valid) if hasattr(g, "compacted_values") else g).reshape(-1, chunks, S)


def _two_rank_worker(rank, world, port, mode, out):
    """Both ranks share cuda:0; the collective transport is gloo (RCCL refuses two ranks on one device), everything else -- the HIP
    mark / compact+rank / scatter primitives, the union-packed buffer, the sizing feedback, Adam over the union -- is the product path."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from litegs_amd import dp, synthetic as S
        from litegs_amd.trainer import SyntheticTrainer
        scene = S.make_scene(20000, seed=4)                             # identical replicas
        tr = SyntheticTrainer(20000, 320, 200, 700.0, n_frames=2 * world, scene=scene)     # narrow field of view: frames see different chunks
        ex = dp.GradientExchange(tr.params, world, mode=mode)
        checks = {}
        # step 0 by hand: the exchanged gradient must be the mean of the two ranks' own dense gradients, over the union of their chunks
        fr = tr.frames[dp.frame_for(0, rank, world, len(tr.frames))]
        img, vis_id, vis_num, _ = tr.forward(fr)
        img.sum().backward()
        n_local = int(vis_num.item())
        local_dense = torch.cat([p.grad.to_dense(n_local).reshape(-1, tr.n_chunks, tr.S) for p in tr.params]).cpu()
        local_mask = torch.zeros(tr.n_chunks, dtype=torch.int32)
        local_mask[vis_id[:n_local].cpu()] = 1
        uid, ucnt = ex.hook(tr.params, vis_id, vis_num, 0)
        gathered = [torch.zeros_like(local_dense) for _ in range(world)]
        dist.all_gather(gathered, local_dense)
        masks = [torch.zeros_like(local_mask) for _ in range(world)]
        dist.all_gather(masks, local_mask)
        union = torch.nonzero(sum(masks))[:, 0]
        U = int(ucnt.item())
        checks['union'] = U == len(union) and torch.equal(uid[:U].cpu(), union)
        checks['differs'] = len(union) > n_local                        # the two frames really see different chunks
        got = torch.cat([_dense_grad(p, U, tr.n_chunks, tr.S) for p in tr.params]).cpu()
        expect = sum(gathered) / world
        checks['mean_grad'] = torch.allclose(got, expect, rtol=1e-5, atol=1e-7 * expect.abs().max().item())
        checks['max_err'] = float((got - expect).abs().max() / expect.abs().max())
        tr.opt.zero_grad(set_to_none=True)
        # then a few full steps: replicas must stay bit-identical
        for i in range(4):
            tr.step(dp.frame_for(i, rank, world, len(tr.frames)), ex.hook, i % 2)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in tr.params]).cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        checks['replicas_identical'] = torch.equal(both[0], both[1])
        checks['moved'] = not torch.equal(flat, torch.cat([torch.from_numpy(a).reshape(-1) for a in scene]))
        out[rank] = checks
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sparse", "dense"])
def test_two_ranks_share_one_gpu_over_gloo(mode):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    for r in range(world):
        c = dict(out.get(r) or {})
        assert c and all(v for k, v in c.items() if k != "max_err"), (r, c)


def _two_rank_epochs_worker(rank, world, port, mode, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from litegs_amd import densify as D, dp
        from litegs_amd.statistics import STATS
        from litegs_amd.trainer import SyntheticTrainer, train
        tr = SyntheticTrainer(8192, 320, 240, 300.0, n_frames=8, seed=2)
        n0 = tr.n_chunks
        tr.enable_densify(D.DensifyParams(densify_from=1, densification_interval=2, opacity_reset_interval=4, target_primitives=20000,
                                          prune_mode="threshold"), total_epochs=10, seed=1)
        ex = dp.MomentExchange(tr.params, world) if mode == "moments" else dp.GradientExchange(tr.params, world)
        sizes = []
        train(tr, 6, ex, rank, world, on_epoch=lambda e, t: sizes.append(t.n_chunks))
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in tr.params]).cpu()
        n = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(n, torch.tensor([flat.numel()]))
        checks = {"same_size": all(int(x) == flat.numel() for x in n), "grew": sizes[2] != n0 and sizes[1] == n0,
                  "stats_follow": STATS.chunks == tr.n_chunks, "finite": bool(torch.isfinite(flat).all())}
        if checks["same_size"]:
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            checks["replicas_identical"] = torch.equal(both[0], both[1])
        out[rank] = checks
    finally:
        dist.destroy_process_group()


@pytest.mark.stochastic
@pytest.mark.parametrize("mode", ["moments", "sparse"])
def test_two_ranks_epochs_with_density_control_stay_identical(mode):
    """six epochs of the data-parallel loop on two ranks (one GPU, gloo transport): statistic epochs, density control (append + prune +
    opacity reset), Morton re-sort, exchange buffers re-bound to the new chunk count -- and the replicas never diverge"""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_epochs_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    for r in range(world):
        c = dict(out.get(r) or {})
        assert c and all(c.values()), (r, c)


def _two_rank_speculation_worker(rank, world, port, out):
    """two ranks on cuda:0 over gloo: data-parallel steps under rank-consistent speculative culling against the gated repeat"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from litegs_amd import dp, synthetic as S
        from litegs_amd.trainer import SyntheticTrainer
        scene = S.make_scene(150_000, seed=5)
        nf, steps = 2 * world, 14

        def run(speculative, drop_step=-1, tight=False):
            tr = SyntheticTrainer(150_000, 640, 360, 380.0, n_frames=nf, scene=scene)
            tr.speculative = speculative
            ex = dp.MomentExchange(tr.params, world, n_slots=2)
            if tight:
                ex.spec_cap_factor = ex.cap_factor = 1.0       # any growth of a slot's record count is an overflow: replayed, exactly
                ex.cap_margin = 0
            rd = tr.renderer
            losses = []
            for i in range(steps):
                peers = [dp.frame_for(i, r, world, nf) for r in range(world)]
                if i in (6, 9) and rank == 0:                  # RANK 0's frame of this step was visited before: sabotage its bounds; rank 1 is fine
                    torch.cuda.synchronize()
                    k = peers[0]
                    gx, gy = -(-640 // 16), -(-360 // 8)
                    upper = sum((-(-gx // (1 << q))) * (-(-gy // (1 << q))) for q in range(1, 4))
                    rd.sched[k, rd.sched_cur[k]][:upper + gx * gy].view(torch.float32).mul_(0.2)
                if i == drop_step:
                    continue
                tr.step(peers[rank], ex, i % 2, peers)
                losses.append(tr.last["loss"])
            tr.flush()
            ex.check()
            torch.cuda.synchronize()
            return tr, ex, [float(l) for l in losses]

        ta, _, la = run(False)
        t_lost, _, _ = run(False, drop_step=12)
        tb, exb, lb = run(True)
        tc, exc, _ = run(True, tight=True)
        c = {}
        c["gated_run_fell_back"] = ta.renderer.fallbacks >= 1 if rank == 0 else True
        c["replays"] = tb.spec_replays
        c["replayed"] = tb.spec_replays >= 2
        c["poison_clear"] = int(tb.renderer.spec_poison.item()) == 0 and int(tc.renderer.spec_poison.item()) == 0
        c["applied_all"] = tb.renderer.applied_step() == steps and tc.renderer.applied_step() >= steps
        c["slot_clean"] = int(exb.slot.abs().sum().item()) == 0 and int(exc.slot.abs().sum().item()) == 0
        c["overflow_replayed"] = exc.overflow_replays >= 1
        c["overflow_replays"] = exc.overflow_replays
        ratios = []
        for pa, pl, pb, pc in zip(ta.params, t_lost.params, tb.params, tc.params):
            d_lost = (pa.detach() - pl.detach()).abs().mean().item()
            ratios.append(max((pa.detach() - pb.detach()).abs().mean().item(), (pa.detach() - pc.detach()).abs().mean().item()) / max(d_lost, 1e-30))
        c["ratio"] = max(ratios)
        c["matches_gated"] = max(ratios) <= 0.06             # the bound of tests/test_gpu_cull.py: geometric middle of noise and a lost step
        c["losses"] = bool(np.allclose(la[:6], lb[:6], rtol=1e-4))       # before the first sabotage the two runs are the same computation
        for name, t in (("spec", tb), ("tight", tc)):
            flat = torch.cat([p.detach().reshape(-1) for p in t.params]).cpu()
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            c["replicas_identical_" + name] = torch.equal(both[0], both[1])
        counts = [None] * world
        dist.all_gather_object(counts, (tb.spec_replays, tc.spec_replays, exc.overflow_replays))
        c["same_replays_on_every_rank"] = counts[0] == counts[1]
        out[rank] = c
    finally:
        dist.destroy_process_group()


@pytest.mark.stochastic
def test_two_ranks_speculative_culling_replays_in_lock_step():
    """Rank-consistent speculative culling (litegs_amd/dp.py): rank 0's bounds are sabotaged before two of its visits, rank 1's never.  Both
    ranks must notice at the same step, replay the same steps (replay counts equal), end with bit-identical replicas and agree with the
    gated-repeat run up to the atomics-order noise (same yardstick as tests/test_gpu_cull.py).  A second speculative run with a block
    capacity of exactly the previous count turns record-count growth into overflows: replayed with an exact capacity, never an error."""
    import torch.multiprocessing as mp
    from util import noise_log
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_speculation_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        c = dict(out.get(r) or {})
        noise_log(what=f"dp spec vs gated / lost step, rank {r}", ratio=c.get("ratio"), bound=0.06, replays=c.get("replays"),
                  overflow_replays=c.get("overflow_replays"))
        assert c and all(v for k, v in c.items() if k not in ("ratio", "replays", "overflow_replays")), (r, c)
