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
    check(lib().lg_fused_backward_adam(pend["A"], pend["S"], R.H, R.W, fr.view_ptr, fr.proj_ptr, pend["degree"], pend["chunks"], pend["Rr"],
                                       pend["vis_ids"].data_ptr(), pend["vis_num"].data_ptr(), pend["pg"].data_ptr(), None,
                                       *[p.data_ptr() for p in ps], *[m.data_ptr() for m in ms], *[v.data_ptr() for v in vs],
                                       lr6, 0.9, 0.999, float(tr.opt.param_groups[0]["eps"]),
                                       touched.data_ptr() if touched is not None else None, torch.cuda.current_stream().cuda_stream), "backward+adam")
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
    # one more forward + blend backward; its moment records stay pending
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
