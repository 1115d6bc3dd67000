"""Exact skip of no-op Adam updates in the fused backward + Adam (csrc/fused.hip): Gaussians of the visible chunks whose moments are
all zero and whose gradient record is all zero are neither read nor written.  The SAME blend-backward output is applied twice to
cloned optimizer states -- with the flag array and without -- and every parameter and moment must come out bit-identical."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

ORDER = ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]


def _apply(tr, pend, ps, ms, vs, touched):
    from litegs_amd._lib import check, lib
    by = {g["name"]: g for g in tr.opt.param_groups}
    lr6 = (ctypes.c_float * 6)(*[float(by[n]["lr"]) for n in ["xyz", "sh_0", "sh_rest", "opacity", "scale", "rot"]])
    R, fr = tr.renderer, pend["frame"]
    check(lib().lg_fused_backward_adam(None, None, pend["A"], pend["S"], R.H, R.W, fr.view_ptr, fr.proj_ptr, pend["degree"], pend["chunks"], pend["Rr"],
                                       pend["vis_ids"].data_ptr(), pend["vis_num"].data_ptr(), pend["pg"].data_ptr(), None,
                                       *[p.data_ptr() for p in ps], *[m.data_ptr() for m in ms], *[v.data_ptr() for v in vs],
                                       lr6, 0.9, 0.999, float(tr.opt.param_groups[0]["eps"]),
                                       touched.data_ptr() if touched is not None else None,
                                       (pend["ws1"].data_ptr() + lib().lg_fused_alloc_offset(pend["A"] * pend["S"])) if touched is not None else None,
                                       torch.cuda.current_stream().cuda_stream), "backward+adam")
    torch.cuda.synchronize()


@pytest.mark.parametrize("degree", [3, 1])
def test_skipping_untouched_gaussians_is_bit_exact(degree):
    from litegs_amd import loss_hip
    from litegs_amd.trainer import SyntheticTrainer
    tr = SyntheticTrainer(40000, 480, 270, 380.0, n_frames=2, seed=3)
    tr.degree = degree
    for i in range(3):                                   # some Gaussians get a history (non-zero moments), most never do
        tr.step(i % 2)
    flags0 = tr.fadam._touched_flags().clone()
    by = {g["name"]: g["params"][0] for g in tr.opt.param_groups}
    params = [by[n] for n in ORDER]
    # one more forward + blend backward; its moment records stay pending (no gradient replicas: _apply below calls the C entry directly)
    tr.renderer.replicas_enabled = False
    tr.renderer.fuse_optimizer = True
    img, vis_id, vis_num, _ = tr.forward(tr.frames[1], raw=True)
    loss_hip.raster_l1_ssim_loss(img, tr.frames[1].gt).backward()
    pend = tr.renderer.pending
    assert pend is not None
    tr.renderer.pending = None
    out = {}
    for mode in ("skip", "full"):
        ps = [p.detach().clone() for p in params]
        ms = [tr.opt.state[p]["exp_avg"].clone() for p in params]
        vs = [tr.opt.state[p]["exp_avg_sq"].clone() for p in params]
        touched = flags0.clone() if mode == "skip" else None
        _apply(tr, pend, ps, ms, vs, touched)
        out[mode] = (ps, ms, vs, touched)
    for group in range(3):
        for a, b in zip(out["skip"][group], out["full"][group]):
            assert torch.equal(a, b)                                     # bit-identical parameters and moments
    flags1 = out["skip"][3]
    n = flags0.numel()
    assert int((flags1 < flags0).sum()) == 0                             # flags only ever go up
    nz = torch.zeros((n,), dtype=torch.bool, device=flags1.device)
    for t in out["full"][1] + out["full"][2]:
        nz |= (t.reshape(-1, n) != 0).any(dim=0)
    assert int((nz & (flags1 == 0)).sum()) == 0                          # flag 0 => every moment of the Gaussian is zero
    vis = int(vis_num.item()) * tr.S
    skipped = vis - int(flags1.view(tr.n_chunks, tr.S)[vis_id[: int(vis_num.item())]].sum())
    assert skipped > 0.3 * vis, (skipped, vis)                           # the case is not vacuous: a large part really was skipped
    assert any(not torch.equal(a, p.detach()) for a, p in zip(out["skip"][0], params))    # and something was updated


def test_flags_are_rebuilt_after_other_optimizer_paths():
    """the gradient-hook path (dense Adam kernel) and density control write moments without the flags: the array is dropped and
    rebuilt from the moments before the next fused step"""
    from litegs_amd.trainer import SyntheticTrainer
    tr = SyntheticTrainer(20000, 320, 200, 300.0, n_frames=2, seed=4)
    tr.step(0)
    assert tr.fadam.touched is not None
    seen = {}
    tr.step(1, grad_hook=lambda params, vis_id, vis_num, slot: seen.setdefault("hook", (vis_id, vis_num)) or (vis_id, vis_num))
    assert "hook" in seen and tr.fadam.touched is None
    tr.step(0)
    f = tr.fadam.touched
    assert f is not None
    n = f.numel()
    nz = torch.zeros((n,), dtype=torch.bool, device=f.device)
    for g in tr.opt.param_groups:
        st = tr.opt.state[g["params"][0]]
        nz |= (st["exp_avg"].reshape(-1, n) != 0).any(dim=0) | (st["exp_avg_sq"].reshape(-1, n) != 0).any(dim=0)
    assert int((nz & (f == 0)).sum()) == 0


@pytest.mark.parametrize("binding", ["ctypes", "ext"])
def test_operator_adam_noop_skip_is_exact(binding):
    """`adamUpdate` (the drop-in operator, stateless): quads whose gradient and both moments are all zero are left untouched -- the
    result equals the plain update bit for bit, including quads where only ONE of the twelve words is non-zero, a -0.0 and a NaN"""
    import numpy as np
    from litegs_amd import fused
    if binding == "ext":
        from litegs_amd.binding import compiled
        if compiled is None:
            pytest.skip("compiled litegs_fused extension not built")
        F = compiled
    else:
        F = fused
    rng = np.random.default_rng(21)
    E, chunks, S, A, nvis = 7, 20, 128, 12, 9
    p = rng.standard_normal((E, chunks, S)).astype(np.float32)
    m = np.zeros((E, chunks, S), np.float32)
    v = np.zeros((E, chunks, S), np.float32)
    g = np.zeros((E, A, S), np.float32)
    ids = rng.permutation(chunks)[:A].astype(np.int64)
    # sprinkle history / gradients over ~10 % of the quads, one word at a time
    for arr in (m, v):
        k = rng.random(arr.shape) < 0.03
        arr[k] = rng.standard_normal(int(k.sum())).astype(np.float32) ** 2 * 1e-3
    k = rng.random(g.shape) < 0.03
    g[k] = rng.standard_normal(int(k.sum())).astype(np.float32)
    g[0, 0, 0] = -0.0                                   # signed zero is still zero
    g[1, 1, 5] = np.nan                                 # NaN is a gradient
    t = lambda a: torch.from_numpy(a.copy()).cuda()    # noqa: E731
    pd, md, vd, gd = t(p), t(m), t(v), t(g)
    F.adamUpdate(pd, gd, md, vd, t(ids), torch.tensor([nvis], dtype=torch.int32).cuda(), 1e-2, 0.9, 0.999, 1e-15)
    # plain update in torch, same operation order as the kernel (fp32): m' = b1*m + (1-b1)*g ; v' = b2*v + (1-b2)*g*g ; p' = p + -lr*m'/(sqrt(v')+eps)
    pt, mt, vt, gt = t(p), t(m), t(v), t(g)
    sel = t(ids)[:nvis]
    b1, b2, lr, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-2), np.float32(1e-15)
    mm = float(b1) * mt[:, sel] + float(np.float32(1.0) - b1) * gt[:, :nvis]
    vv = float(b2) * vt[:, sel] + float(np.float32(1.0) - b2) * gt[:, :nvis] * gt[:, :nvis]
    pt[:, sel] = pt[:, sel] + (-float(lr)) * mm / (vv.sqrt() + float(eps))
    mt[:, sel], vt[:, sel] = mm, vv
    noop = ((gt[:, :nvis] == 0) & (mt[:, sel] == 0) & (vt[:, sel] == 0))
    assert float(noop.float().mean()) > 0.8
    for got, want, name in ((pd, pt, "param"), (md, mt, "m"), (vd, vt, "v")):
        assert torch.equal(got[:, sel][noop], want[:, sel][noop]), name                      # untouched where it is a no-op
        a, b = got[:, sel][~noop], want[:, sel][~noop]
        fin = torch.isfinite(b)
        assert torch.equal(torch.isnan(a), torch.isnan(b)), name
        assert torch.allclose(a[fin], b[fin], rtol=2e-6, atol=1e-7), name
    rest = torch.ones(chunks, dtype=torch.bool, device="cuda"); rest[sel] = False
    assert torch.equal(pd[:, rest], t(p)[:, rest]) and torch.equal(md[:, rest], t(m)[:, rest])      # invisible chunks untouched
