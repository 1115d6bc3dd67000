"""world_size=2 data-parallel exchange on CPU (gloo): union visibility, one averaged buffer packed over the union, replicas stay identical.
The device primitives are injected as plain-torch ops (the product's HipOps need a GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class TorchOps:
    @staticmethod
    def mark(mask, ids, count):
        mask[ids[: int(count)]] = 1

    @staticmethod
    def compact(mask):
        ids = torch.nonzero(mask)[:, 0]
        full = torch.arange(mask.shape[0], dtype=torch.int64)
        full[: len(ids)] = ids
        rank = torch.zeros(mask.shape[0], dtype=torch.int64)
        rank[ids] = torch.arange(len(ids))
        return full, torch.tensor([len(ids)], dtype=torch.int32), rank

    @staticmethod
    def scatter_add(dense, compact, ids, count):
        n = int(count)
        keep = ids[:n] < dense.shape[1]
        dense[:, ids[:n][keep], :] += compact[:, :n, :][:, keep]

    @staticmethod
    def feedback(host_slot, count):
        host_slot.copy_(count)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd import dp
    from litegs_amd.wrapper import CompactedTensor
    chunks, S = 12, 4
    g = torch.Generator().manual_seed(0)
    shapes = [(3, chunks, S), (3, chunks, S), (4, chunks, S), (1, 3, chunks, S), (15, 3, chunks, S), (1, chunks, S)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]            # identical replicas
    ex = dp.GradientExchange(params, world, ops=TorchOps, mode="dense")
    # rank-specific visibility (over-allocated id list, as with the 1.2x prediction) and compact gradients
    vis = [torch.tensor([1, 4, 5, 9, 0, 0]), torch.tensor([4, 5, 6, 11, 2, 0])][rank]
    cnt = torch.tensor([4], dtype=torch.int32)
    gr = torch.Generator().manual_seed(100 + rank)
    dense_local = []
    for p in params:
        rows = p.numel() // (chunks * S)
        vals = torch.randn((rows, len(vis), S), generator=gr)
        p.grad = CompactedTensor(p.shape, vis, vals)
        d = torch.zeros(rows, chunks, S)
        d[:, vis[:4]] = vals[:, :4]
        dense_local.append(d)
    gathered = [torch.zeros(sum(ex.rows), chunks, S) for _ in range(world)]
    dist.all_gather(gathered, torch.cat(dense_local))
    expect = sum(gathered) / world
    ok = True
    for visit in range(2):                 # visit 0: blocking exact size; visit 1: size predicted from the feedback slot (1.2x + 1)
        for p, d in zip(params, dense_local):
            rows = p.numel() // (chunks * S)
            p.grad = CompactedTensor(p.shape, vis, d[:, vis].clone())
        union_ids, union_count = ex.hook(params, vis, cnt, slot=3)
        got = torch.cat([p.grad.to_dense(int(union_count)).reshape(-1, chunks, S) for p in params])
        ok &= torch.allclose(got, expect, atol=1e-6)
        ok &= int(union_count) == 6
        ok &= union_ids[: int(union_count)].tolist() == [1, 4, 5, 6, 9, 11]
        ok &= all(p.grad.shape == p.shape and hasattr(p.grad, "compacted_values") for p in params)
        ok &= ex.last_alloc == (6 if visit == 0 else 8) and int(ex.fb_union[3]) == 6
        ok &= all(p.grad.compacted_values.shape[-2] == ex.last_alloc for p in params)
    ok &= dp.frame_for(3, rank, world, 8) == (3 * world + rank) % 8
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_gradient_exchange_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _stats_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd.statistics import Statistics, _Moments
    chunks, S = 5, 4
    st = Statistics()
    st.chunks, st.S, st.device = chunks, S, "cpu"
    g = torch.Generator().manual_seed(rank)
    st.visible_count = torch.randint(0, 3, (chunks, S), generator=g, dtype=torch.int32)
    mom = _Moments((1,), chunks, S, "cpu")
    mom.sum = torch.rand((1, chunks, S), generator=g)
    mom.square_sum = torch.rand((1, chunks, S), generator=g)
    mom.count = torch.randint(0, 5, (chunks, S), generator=g, dtype=torch.int32)
    st.moments["fragment_err"] = mom
    local = (st.visible_count.clone(), mom.sum.clone(), mom.square_sum.clone(), mom.count.clone())
    st.all_reduce()
    ok = True
    for mine, reduced in zip(local, (st.visible_count, mom.sum, mom.square_sum, mom.count)):
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        ok &= torch.allclose(sum(parts).double(), reduced.double())
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_statistics_all_reduce_world2():
    """densification inputs must be identical on every rank: sums / squared sums / counts are summed across the job"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_stats_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _densify_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd import densify as D
    from litegs_amd import optimizer as opt_mod
    from litegs_amd.statistics import Statistics, _Moments
    chunks, S = 6, 32
    g = torch.Generator().manual_seed(5)                                  # identical replicas
    shapes = [(3, chunks, S), (3, chunks, S), (4, chunks, S), (1, 3, chunks, S), (3, 3, chunks, S), (1, chunks, S)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    with torch.no_grad():
        ps[1].mul_(0.5).sub_(3.0)
    opt, _ = opt_mod.get_optimizer(*ps, 1.0, opt_mod.OptimizationParams())
    for grp in opt.param_groups:
        p = grp["params"][0]
        opt.state[p] = {"step": torch.tensor(1.0), "exp_avg": torch.randn(p.shape, generator=g), "exp_avg_sq": torch.rand(p.shape, generator=g)}
    # every rank rendered DIFFERENT frames: rank-specific evidence
    st = Statistics()
    st.reset(chunks, S, None, device="cpu")
    gr = torch.Generator().manual_seed(50 + rank)
    for key in ("fragment_err", "fragment_weight"):
        mom = _Moments((1,), chunks, S, "cpu")
        mom.sum = torch.rand((1, chunks, S), generator=gr)
        mom.square_sum = mom.sum ** 2 + torch.rand((1, chunks, S), generator=gr)
        mom.count = torch.randint(0, 3, (chunks, S), generator=gr, dtype=torch.int32)
        st.moments[key] = mom
    dp_ = D.DensifyParams(densify_until=41, target_primitives=2000)
    ctl = D.DensityController(2.0, dp_, S, chunks * S, st, D.Sampler(11))
    new = ctl.step(opt, 5)
    flat = torch.cat([t.detach().reshape(-1) for t in new] + [opt.state[t]["exp_avg"].reshape(-1) for t in new])
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([flat.numel()]))
    same_size = all(int(s) == flat.numel() for s in sizes)
    ok = same_size
    if same_size:
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        ok = all(torch.equal(both[0], b) for b in both) and new[0].shape[-2] != chunks and st.chunks == new[0].shape[-2]
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_density_control_is_identical_on_every_rank_world2():
    """rank-specific statistics are summed first and the random draws depend on (seed, epoch) only: after a densification step the
    replicas hold bit-identical parameters and Adam moments (SURVEY 8f-1 "deterministic densify")"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_densify_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _sparse_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd import dp
    from litegs_amd.wrapper import CompactedTensor
    chunks, S = 12, 4
    g = torch.Generator().manual_seed(0)
    shapes = [(3, chunks, S), (3, chunks, S), (4, chunks, S), (1, 3, chunks, S), (15, 3, chunks, S), (1, chunks, S)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    ex = dp.GradientExchange(params, world, ops=TorchOps)                 # default mode
    assert ex.mode == "sparse"
    vis = [torch.tensor([1, 4, 5, 9, 0, 0]), torch.tensor([4, 5, 6, 11, 2, 0])][rank]      # over-allocated id lists, 4 valid each
    cnt = torch.tensor([4], dtype=torch.int32)
    gr = torch.Generator().manual_seed(100 + rank)
    # which Gaussians of the visible chunks got a gradient at all (about a third); the rest are exact zeros
    touched = torch.rand((len(vis), S), generator=gr) < 0.35
    touched[4:] = True                                                     # dirty tail beyond the valid count: must be ignored
    dense_local, compact = [], []
    for p in params:
        rows = p.numel() // (chunks * S)
        vals = torch.randn((rows, len(vis), S), generator=gr) * touched
        vals[:, 4:] = 7.0                                                  # garbage in the tail
        compact.append(vals)
        d = torch.zeros(rows, chunks, S)
        d[:, vis[:4]] = vals[:, :4]
        dense_local.append(d)
    gathered = [torch.zeros(sum(ex.rows), chunks, S) for _ in range(world)]
    dist.all_gather(gathered, torch.cat(dense_local))
    expect = sum(gathered) / world
    ok = True
    for visit in range(2):
        for p, v in zip(params, compact):
            p.grad = CompactedTensor(p.shape, vis, v.clone())
        union_ids, union_count = ex.hook(params, vis, cnt, slot=visit)
        got = torch.cat([p.grad.reshape(-1, chunks, S) for p in params])
        ok &= torch.allclose(got, expect, atol=1e-6)
        ok &= all(type(p.grad) is torch.Tensor and p.grad.shape == p.shape for p in params)
        ok &= int(union_count) == 6 and union_ids[:6].tolist() == [1, 4, 5, 6, 9, 11]
        ok &= ex.last_k[0] == int(touched[:4].sum()) and ex.last_k[1] >= ex.last_k[0]
    # bit-identical on every rank (rank-ordered accumulation), not merely close
    both = [torch.zeros_like(got) for _ in range(world)]
    dist.all_gather(both, got)
    ok &= torch.equal(both[0], both[1])
    # a rank that saw nothing still takes part
    for p, v in zip(params, compact):
        p.grad = CompactedTensor(p.shape, vis, v.clone())
    zero_cnt = torch.tensor([0 if rank == 1 else 4], dtype=torch.int32)
    union_ids, union_count = ex.hook(params, vis, zero_cnt, slot=0)
    got = torch.cat([p.grad.reshape(-1, chunks, S) for p in params])
    ok &= torch.allclose(got, gathered[0] / world, atol=1e-6) and int(union_count) == 4
    # after a re-sort / density control the exchange is re-bound: the gradient buffer must be clean again
    ex.rebind(params)
    ok &= float(ex.flat.abs().max()) == 0.0
    for p, v in zip(params, compact):
        p.grad = CompactedTensor(p.shape, vis, v.clone())
    ex.hook(params, vis, cnt, slot=1)
    got = torch.cat([p.grad.reshape(-1, chunks, S) for p in params])
    ok &= torch.allclose(got, expect, atol=1e-6)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sparse_gradient_exchange_world2():
    """default exchange: only Gaussians with a non-zero gradient travel (all_gather of [values | index] blocks sized by the job's
    largest count), accumulated in rank order into a dense gradient: equals the mean of the dense gradients, identical on all ranks"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sparse_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


# ------------------------------------------------------------------------------------------------------------------------------
# mode "moments" (litegs_amd/dp.py: MomentExchange): the collective / sizing / slot-map logic on CPU with plain-torch primitives.
# The stand-in for the fused backward + Adam kernel just accumulates the records it is handed, in rank order, into ps[0].
# ------------------------------------------------------------------------------------------------------------------------------
class TorchMomentOps:
    NREC = 10

    @staticmethod
    def compact_moments(pg, vis_ids, vis_num, A, S, cap, block):
        n = int(vis_num) * S
        rows = pg[:n, :9]
        nz = (rows != 0).any(dim=1).nonzero()[:, 0]
        blk = block.view(torch.float32).view(-1, TorchMomentOps.NREC)          # the wire container is int32 (litegs_amd/dp.py)
        blk[0].zero_()
        blk[0, :1].view(torch.int32)[0] = len(nz)
        keep = nz[:cap]
        gid = (vis_ids[keep // S] * S + keep % S).to(torch.int32)
        blk[1:1 + len(keep), 0] = gid.view(torch.float32)
        blk[1:1 + len(keep), 1:] = rows[keep]

    @staticmethod
    def build_slotmap(gathered, W, cap, total, slot, host_max_k_ptr, overflow):
        g = gathered.view(torch.float32).view(W, 1 + cap, TorchMomentOps.NREC)
        ks = [int(g[r, 0, :1].view(torch.int32)[0]) for r in range(W)]
        for r in range(W):
            k = min(ks[r], cap)
            gid = g[r, 1:1 + k, 0].contiguous().view(torch.int32).long()
            slot[r, gid] = torch.arange(1, k + 1, dtype=torch.int32)
        TorchMomentOps.fb_target[0] = max(ks)                 # the two pinned words of csrc/dp.hip: {largest count, overflow marker}
        TorchMomentOps.fb_target[1] = max(ks) if max(ks) > cap else 0
        if max(ks) > cap:
            overflow[0] = 1

    @staticmethod
    def backward_adam(union_ids, union_count, chunks, S, H, Wimg, views, projs, W, degree, R, gathered, cap, slot, ps, ms, vs, lr6, eps, touched=None):
        g = gathered.view(torch.float32).view(W, 1 + cap, TorchMomentOps.NREC)
        acc = ps[0].view(9, chunks * S)
        acc.zero_()
        for c in union_ids[: int(union_count)].tolist():
            for t in range(S):
                gid = c * S + t
                for r in range(W):                     # rank order
                    k1 = int(slot[r, gid])
                    if k1:
                        slot[r, gid] = 0
                        acc[:, gid] += g[r, k1, 1:] * float(views[r][0])        # "camera" r scales rank r's records
        acc /= W


def _moment_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd import dp
    chunks, S = 10, 4
    params = [torch.zeros((9, chunks, S))]
    ex = dp.MomentExchange(params, world, ops=TorchMomentOps, union_ops=TorchOps)
    ex.cap_margin = 1
    vis = [torch.tensor([1, 4, 5, 9, 0, 0]), torch.tensor([4, 5, 6, 8, 2, 0])][rank]
    cnt = torch.tensor([4 if rank == 0 else 5], dtype=torch.int32)
    A = len(vis)
    ok = True
    cams = [([2.0] + [0.0] * 15, [0.0] * 16), ([3.0] + [0.0] * 15, [0.0] * 16)]
    for visit in range(3):
        g = torch.Generator().manual_seed(10 * visit + rank)
        pg = torch.zeros((A * S, 16))
        touched = torch.rand((A * S,), generator=g) < (0.5 if visit < 2 else 0.95)          # visit 2: many more records than predicted
        if rank == 1 and visit == 1:
            touched[:] = False                                                               # a rank that touched nothing
        pg[touched, :9] = torch.randn((int(touched.sum()), 9), generator=g)
        pg[int(cnt) * S:] = 7.0                                                              # slots beyond vis_num are dirty by design
        dense = torch.zeros((9, chunks * S))
        n = int(cnt) * S
        gid = (vis[torch.arange(n) // S] * S + torch.arange(n) % S)
        dense[:, gid] = pg[:n, :9].t() * cams[rank][0][0]
        both = [torch.zeros_like(dense) for _ in range(world)]
        dist.all_gather(both, dense)
        expect = (both[0] + both[1]) / world
        TorchMomentOps.fb_target = ex.fb_k[3]
        pend = dict(pg=pg, A=A, S=S, vis_ids=vis, vis_num=cnt, degree=3, Rr=15)
        if visit == 1:                                   # the asynchronous form: the union collective is started early (the renderer does
            ex.begin(vis, cnt)                           # this right after the culling), step() only waits for it
            ok &= ex._mask_work is not None
        uid, ucnt = ex.step(pend, cams, params, [None], [None], [0.0] * 6, 1e-15, 8, 8, slot=3)
        ok &= ex._mask_work is None
        why = []
        if uid[: int(ucnt)].tolist() != [1, 2, 4, 5, 6, 8, 9]:
            why.append(("union", uid[: int(ucnt)].tolist()))
        got = params[0].view(9, chunks * S)
        if visit < 2:
            if not torch.allclose(got, expect, atol=1e-6):
                why.append(("values", visit, float((got - expect).abs().max())))
            if int(ex.overflow[0]) != 0:
                why.append(("overflow", visit, ex.last_cap))
            ok &= not why
            if why:
                out[f"why{rank}"] = str(why)
        else:                                            # capacity predicted from visit 1 is outgrown: loud failure, not silence
            try:
                ex.check()
                ok = False
            except RuntimeError as e:
                ok &= "step 3" in str(e)                 # the step that dropped records is named
        ok &= int(ex.slot.abs().sum()) == 0              # the map is left clean
        ok &= int(ex.fb_k[3, 0]) > 0
    try:                                                 # a slot beyond n_slots must not alias another frame set's prediction
        ex.step(pend, cams, params, [None], [None], [0.0] * 6, 1e-15, 8, 8, slot=ex.n_slots)
        ok = False
    except ValueError:
        pass
    ex.ensure_slots(ex.n_slots + 3)
    ok &= ex.fb_k.shape[0] == ex.n_slots and int(ex.fb_k[3, 0]) > 0
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_moment_exchange_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_moment_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


# ------------------------------------------------------------------------------------------------------------------------------
# Rank-consistent speculative culling (litegs_amd/dp.py: Speculation, LockstepSpeculation, MomentExchange.enable_speculation): the
# verdict on a step is a function of the gathered headers, the hosts read it at a fixed lag and replay in lock-step.  The device
# is played by plain-torch primitives that honour a poison word exactly as csrc/dp.hip does.
# ------------------------------------------------------------------------------------------------------------------------------
class TorchSpecMomentOps(TorchMomentOps):
    @staticmethod
    def compact_moments(pg, vis_ids, vis_num, A, S, cap, block, spec=None):
        TorchMomentOps.compact_moments(pg, vis_ids, vis_num, A, S, cap, block)
        if spec is not None:
            block.view(-1, TorchMomentOps.NREC)[0, 1] = 1 if int(spec.poison[0]) != 0 else 0

    @staticmethod
    def build_slotmap(gathered, W, cap, total, slot, host_max_k_ptr, overflow, spec=None):
        if spec is None:
            return TorchMomentOps.build_slotmap(gathered, W, cap, total, slot, host_max_k_ptr, overflow)
        g = gathered.view(W, 1 + cap, TorchMomentOps.NREC)
        ks = [int(g[r, 0, 0]) for r in range(W)]
        for r in range(W):
            k = min(ks[r], cap)
            slot[r, g[r, 1:1 + k, 0].long()] = torch.arange(1, k + 1, dtype=torch.int32)
        flags = 0
        for r in range(W):
            flags |= (1 << r) if int(g[r, 0, 1]) != 0 else 0
        if max(ks) > cap:
            flags |= 1 << 8
        if flags:
            spec.poison[0] = 1
        TorchMomentOps.fb_target[0] = max(ks)
        TorchMomentOps.fb_target[1] = 0
        i = spec.status_addr(spec.step_id)
        spec.words[i + 1] = flags
        spec.words[i] = spec.step_id

    @staticmethod
    def backward_adam(union_ids, union_count, chunks, S, H, Wimg, views, projs, W, degree, R, gathered, cap, slot, ps, ms, vs, lr6, eps, touched=None,
                      spec=None):
        if spec is not None and int(spec.poison[0]) != 0:
            return                                         # a failed step: nothing changes (the slot map stays dirty, as on the device)
        g = gathered.view(torch.float32).view(W, 1 + cap, TorchMomentOps.NREC)
        acc = torch.zeros((9, chunks * S))
        for r in range(W):                                 # rank order
            for gid in slot[r].nonzero()[:, 0].tolist():
                acc[:, gid] += g[r, int(slot[r, gid]), 1:] * float(views[r][0])
        slot.zero_()
        ps[0].view(9, chunks * S).add_(acc / W * lr6[0])   # "Adam": the learning rate of the step scales the update
        if spec is not None:
            TorchSpecMomentOps.applied.append(spec.step_id)


def _spec_worker(rank, world, port, out):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from litegs_amd import dp
    chunks, S, T = 10, 4, 14
    vis = [torch.tensor([1, 4, 5, 9, 0, 0]), torch.tensor([4, 5, 6, 8, 2, 0])][rank]
    cnt = torch.tensor([4 if rank == 0 else 5], dtype=torch.int32)
    A = len(vis)
    cams = [([2.0] + [0.0] * 15, [0.0] * 16), ([3.0] + [0.0] * 15, [0.0] * 16)]
    fail_plan = {(5, 1), (6, 0), (11, 0), (11, 1)}          # (step, rank): that rank's culled forward violates a bound (unless unculled)
    dense_step = 9                                          # ... and here rank 1 produces far more records than the slot's last visit

    def records(no, r):
        g = torch.Generator().manual_seed(100 * no + r)
        pg = torch.zeros((A * S, 16))
        touched = torch.rand((A * S,), generator=g) < (0.95 if (no == dense_step and r == 1) else 0.5)
        pg[touched, :9] = torch.randn((int(touched.sum()), 9), generator=g)
        pg[int([4, 5][r]) * S:] = 7.0                       # slots beyond vis_num are dirty by design
        return pg

    def lr_of(no):
        return 1.0 / no                                     # a schedule: a replayed step must run with ITS learning rate

    # what T steps must add up to, each applied exactly once (both ranks' records, rank order, mean)
    expect = torch.zeros((9, chunks * S))
    for no in range(1, T + 1):
        for r in range(world):
            v, n = [torch.tensor([1, 4, 5, 9, 0, 0]), torch.tensor([4, 5, 6, 8, 2, 0])][r], [4, 5][r] * S
            gid = v[torch.arange(n) // S] * S + torch.arange(n) % S
            expect[:, gid] += records(no, r)[:n, :9].t() * cams[r][0][0] / world * lr_of(no)

    params = [torch.zeros((9, chunks, S))]
    ex = dp.MomentExchange(params, world, ops=TorchSpecMomentOps, union_ops=TorchOps, n_slots=4)
    ex.cap_margin = 1
    poison = torch.zeros((1,), dtype=torch.int32)
    ex.enable_speculation(poison, None, words=np.zeros((2 * dp.Speculation.RING,), dtype=np.int32))
    TorchSpecMomentOps.applied = []
    log = []

    def body(rec, force):
        no, slot = rec
        log.append((no, bool(force)))
        if (no, rank) in fail_plan and not force:
            poison[0] = 1                                   # the culled forward's bound check
        TorchMomentOps.fb_target = ex.fb_k[slot]
        pend = dict(pg=records(no, rank), A=A, S=S, vis_ids=vis, vis_num=cnt, degree=3, Rr=15)
        ex.step(pend, cams, params, [None], [None], [lr_of(no)] + [0.0] * 5, 1e-15, 8, 8, slot=slot, step_id=no)

    class Done:
        def synchronize(self):
            pass

    failed = []

    def on_failed(rec, flags):
        failed.append((rec[0], flags))
        poison[0] = 0
        ex.after_failed_step(rec[1], flags)

    ls = dp.LockstepSpeculation(ex, 2, body, on_failed, Done, lambda: None)
    for no in range(1, T + 1):
        rec = (no, no % 3)
        ls.before_step(rec)
        body(rec, False)
        ls.after_step(rec)
    ls.flush()
    got = params[0].view(9, chunks * S)
    other = [None] * world
    dist.all_gather_object(other, (log, failed, TorchSpecMomentOps.applied))
    why = []
    if not torch.allclose(got, expect, atol=1e-5):
        why.append(("values", float((got - expect).abs().max())))
    if other[0] != other[1]:
        why.append(("ranks took different decisions", other))
    if sorted(TorchSpecMomentOps.applied) != list(range(1, T + 1)):
        why.append(("every step's update exactly once", TorchSpecMomentOps.applied))
    kinds = {no: fl for no, fl in failed}
    # step 5: rank 1 alone; step 6 fails in the replay behind it (rank 0); step 9: overflow; step 11: both ranks
    if not (kinds.get(5, 0) & 0xff == 0b10 and kinds.get(6, 0) & 0xff == 0b01 and kinds.get(9, 0) & (1 << 8) and kinds.get(11, 0) & 0xff == 0b11):
        why.append(("failures", failed))
    if ex.overflow_replays != 1 or int(poison[0]) != 0 or int(ex.slot.abs().sum()) != 0 or int(ex.overflow[0]) != 0:
        why.append(("end state", ex.overflow_replays, int(poison[0]), int(ex.slot.abs().sum())))
    if ls.replays < 6:
        why.append(("replays", ls.replays, log))
    ex.check()                                              # an overflow under speculation is a replayed step, never an error
    out[rank] = not why
    if why:
        out[f"why{rank}"] = str(why)
    dist.destroy_process_group()


def test_speculative_culling_stays_rank_consistent_world2():
    """failed culled forwards on either rank, a failure inside a replay and a record-block overflow: both ranks take the same decisions at
    the same steps (the collectives would deadlock or pair up wrongly otherwise), every step's update lands exactly once, with its own
    learning rate, and the result equals the plain sum"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_spec_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)
