#!/usr/bin/env python
"""The reference's OWN Python layers executed on this backend.

Loads litegs/utils/wrapper.py, litegs/render/__init__.py, litegs/utils/statistic_helper.py, litegs/training/optimizer.py of the
reference UNMODIFIED (from /root/reference, or from a staged copy under _refstage/ on the GPU box -- tools/stage_reference.sh; the
copy is git-ignored and never committed), with `litegs_fused`, `fused_ssim` and `simple_knn` resolving to this repository, then runs
    render_preprocess -> render -> (img * w).sum().backward() -> SparseGaussianAdam.step
through the reference's wrapper classes (MVPTransform, Binning with its unstable torch.sort + int64 point ids, GaussiansRasterFunc
with its grad_rgb_image_max normalisation, CullCompactActivateWithSparseGrad, ...) and compares image and parameter gradients with
the CPU oracle.  Writes profiles/r02_reference_layers.log when run with --log.
"""
import argparse
import importlib
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_reference():
    base = "/root/reference" if os.path.isdir("/root/reference/litegs") else os.path.join(ROOT, "_refstage")
    if not os.path.isdir(os.path.join(base, "litegs")):
        raise SystemExit("reference sources not found (run tools/stage_reference.sh before gpurun)")
    pkg = types.ModuleType("litegs")                       # package skeleton: litegs/__init__.py would pull in I/O packages (plyfile, cv2)
    pkg.__path__ = [os.path.join(base, "litegs")]
    sys.modules["litegs"] = pkg
    for missing in ("cv2", "plyfile"):
        try:
            importlib.import_module(missing)
        except ImportError:
            sys.modules[missing] = types.ModuleType(missing)
    import litegs_fused                                    # this repository's module of that name
    wrapper = importlib.import_module("litegs.utils.wrapper")
    assert wrapper.litegs_fused is litegs_fused
    render = importlib.import_module("litegs.render")
    arguments = importlib.import_module("litegs.arguments")
    tr = types.ModuleType("litegs.training")               # skeleton again: litegs/training/__init__.py imports the trainer (datasets, I/O)
    tr.__path__ = [os.path.join(base, "litegs", "training")]
    sys.modules["litegs.training"] = tr
    optimizer = importlib.import_module("litegs.training.optimizer")
    return base, litegs_fused, wrapper, render, arguments, optimizer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="small")
    ap.add_argument("--log", action="store_true")
    args = ap.parse_args()
    from tests.util import case
    from oracle import oracle as O
    O.build()
    base, litegs_fused, wrapper, render, arguments, optimizer = load_reference()
    lines = []

    def say(msg):
        print(msg, flush=True)
        lines.append(msg)
    say(f"reference layers from {base}; litegs_fused binding: {getattr(litegs_fused, 'BINDING', '?')} ({litegs_fused.__file__})")
    pp = arguments.PipelineParams.__new__(arguments.PipelineParams)          # class attributes carry the defaults (no argparse needed)
    ok = True
    for name in args.case.split(","):
        c = case(name)
        H, W = c["H"], c["W"]
        dev = torch.device("cuda")
        params = [torch.nn.Parameter(torch.from_numpy(p).to(dev)) for p in c["params"]]
        view, proj, planes = [torch.from_numpy(x).to(dev) for x in (c["view"], c["proj"], c["planes"])]
        t0 = time.time()
        vis_id, vis_num, xyz, scale, rot, color, opacity = render.render_preprocess(None, None, planes, view, *params, None, None, pp, c["degree"])
        img, trans, depth, normal, prim_vis = render.render(view, proj, xyz, scale, rot, color, opacity, vis_num * pp.cluster_size, None, None,
                                                            c["degree"], (H, W), pp)
        rng = np.random.default_rng(4)
        w_host = rng.standard_normal((1, 3, H, W)).astype(np.float32)
        (img * torch.from_numpy(w_host).to(dev)).sum().backward()
        torch.cuda.synchronize()
        t_gpu = time.time() - t0
        res = O.render_forward(c["params"], c["view"], c["proj"], c["planes"], H, W, c["degree"])
        ref_img = np.clip(res.img[..., :H, :W], 0, 1)
        err = np.abs(img.detach().cpu().numpy() - ref_img)
        nbad = int((err > 1e-4).sum())
        say(f"[{name}] visible chunks {int(vis_num.item())} (oracle {res.nvis}); image max err {err.max():.3e}, {nbad} of {err.size} pixels beyond 1e-4")
        ok &= int(vis_num.item()) == res.nvis and nbad <= max(2, int(5e-5 * err.size)) and err.max() < 2e-2
        d_img = np.zeros_like(res.img)
        inside = (res.img[..., :H, :W] >= 0) & (res.img[..., :H, :W] <= 1)
        d_img[..., :H, :W] = w_host * inside
        (grads, _) = O.render_backward(res, c["params"], c["view"], c["proj"], d_img, H, W, c["degree"])
        for p, g_ref, nm in zip(params, grads, ["xyz", "scale", "rot", "sh_0", "sh_rest", "opacity"]):
            g = p.grad
            vals = g.compacted_values if hasattr(g, "compacted_values") else g
            got = vals.detach().cpu().numpy()
            if got.shape != g_ref.shape:                       # compact gradient [.., A, S] -> first nvis chunks
                got = got.reshape(g_ref.shape[:-2] + (-1, g_ref.shape[-1]))[..., :res.nvis, :].reshape(g_ref.shape)
            e = np.abs(got - g_ref) / max(np.abs(g_ref).max(), 1e-30)
            nb = int((e > 1e-4).sum())
            say(f"[{name}] grad.{nm}: max normalised err {e.max():.3e}, {nb} of {e.size} beyond 1e-4")
            ok &= nb <= int(np.ceil(1e-3 * e.size)) and e.max() < 5e-2
        # the reference's optimizer on top of the gradients it just produced
        try:
            op = arguments.OptimizationParams.__new__(arguments.OptimizationParams)
            opt, sched = optimizer.get_optimizer(*params, 1.0, op, pp)
            before = [p.detach().clone() for p in params]
            opt.step(vis_id, vis_num, prim_vis)
            torch.cuda.synchronize()
            moved = sum(int((a != b).any().item()) for a, b in zip(before, params))
            say(f"[{name}] reference SparseGaussianAdam.step through litegs_fused.adamUpdate: {moved} of 6 parameter tensors updated")
            ok &= moved == 6
        except Exception as e:  # noqa: BLE001
            say(f"[{name}] reference optimizer step failed: {type(e).__name__}: {e}")
            ok = False
        say(f"[{name}] wall {t_gpu * 1e3:.1f} ms for preprocess + render + backward (first call, includes allocations)")
    say("RESULT: " + ("PASS" if ok else "FAIL"))
    if args.log:
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        with open(os.path.join(ROOT, "profiles", "r02_reference_layers.log"), "w") as f:
            f.write("\n".join(lines) + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
