#!/usr/bin/env python
"""Records how the UNMODIFIED reference Python drives the `litegs_fused` boundary during one training iteration, as a fixture.

Runs HERE (this container: /root/reference present, no GPU).  The reference's own modules -- litegs/utils/wrapper.py, litegs/render/__init__.py,
litegs/training/optimizer.py, litegs/utils/statistic_helper.py, litegs/scene/cluster.py -- are imported as they are; `litegs_fused` is replaced by a
shape-only stand-in (every operator returns tensors of the shape and dtype the reference's module returns: GR/*.cu as summarised in
SURVEY.md 8a; values are zeros / ones, nothing is computed) that RECORDS every call: operator name, and for every argument and result its
dtype, shape and contiguity (tensors), its type and value (ints, bools) or its type (floats).  One iteration is

    render_preprocess -> render -> (img * w).sum().backward() -> SparseGaussianAdam.step

exactly as litegs/training/trainer.py:129-158 strings them together.  The recorded sequence -- which operators, in which order, how many times,
with which dtypes and layouts -- is the reference's call pattern at the boundary; sizes that depend on the data (Gaussians, visible chunks, table
length) are normalised to symbols.  tests/test_gpu_reference_call_pattern.py replays the same iteration through this repository's mirror of those
layers on the GPU and requires the identical sequence (it is then the thing `bench.py` times as `reference_call_pattern_ms`).

The Python reference itself never leaves this container (it cannot travel to the GPU box in any form): only this trace does.

    python tests/golden/make_call_trace.py        -> tests/golden/reference_call_trace.json
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "reference_call_trace.json")

TRACE = []


def _desc(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return {"t": str(x.dtype).replace("torch.", ""), "shape": list(x.shape), "contig": bool(x.is_contiguous()), "dev": x.device.type}
    if isinstance(x, bool):
        return {"bool": x}
    if isinstance(x, int):
        return {"int": x}
    if isinstance(x, float):
        return {"float": None}
    if isinstance(x, (list, tuple)):
        return [_desc(v) for v in x]
    return {"py": type(x).__name__}


def recorded(fn):
    def wrapper(*args, **kwargs):
        out = fn(*args, **kwargs)
        TRACE.append({"op": fn.__name__, "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in kwargs.items()}, "out": _desc(out)})
        return out
    wrapper.__name__ = fn.__name__
    return wrapper


def make_shim():
    """shape-only `litegs_fused` (GR/ext_cuda.cpp:10-35): result shapes and dtypes of the reference's operators"""
    m = types.ModuleType("litegs_fused")
    f32, i32, i64 = torch.float32, torch.int32, torch.int64
    z = lambda *s, dtype=f32: torch.zeros(s, dtype=dtype)

    def tiles_of(h, w, th, tw):
        gy, gx = int(math.ceil(h / th)), int(math.ceil(w / tw))
        return gy * th, gx * tw, gx * gy

    @recorded
    def frustum_culling_aabb(aabb_origin, aabb_ext, frustumplane, feedback, idx):
        M = aabb_origin.shape[1]
        return [torch.ones(M, dtype=torch.bool), torch.full((1,), M, dtype=i32), torch.arange(M, dtype=i64)]

    @recorded
    def cull_compact_activate(deg, chunk_id, num, view, pos, scale, rot, sh0, shr, opa):
        A, S, V = chunk_id.shape[0], pos.shape[-1], view.shape[0]
        return [z(4, A, S), z(3, A, S), z(4, A, S), z(V, 3, A, S), z(1, A, S)]

    @recorded
    def activate_backward(deg, chunk_id, num, view, pos, scale, rot, sh0, shr, opa, g_pos, g_scale, g_rot, g_color, g_opa):
        A, S = chunk_id.shape[0], pos.shape[-1]
        return [z(pos.shape[0], A, S), z(3, A, S), z(4, A, S), z(sh0.shape[0], sh0.shape[1], A, S), z(shr.shape[0], shr.shape[1], A, S), z(1, A, S)]

    @recorded
    def mvp_transform_forward(world, view, proj, valid_length):
        V, N = view.shape[0], world.shape[1]
        vp = z(V, 4, N); vp[:, 2] = torch.arange(N, dtype=f32) + 1.0        # distinct positive depths
        return [vp, z(V, 4, N)]

    @recorded
    def mvp_transform_backward(g_ndc, g_view, view, proj, view_pos, valid_length):
        return z(4, g_ndc.shape[2])

    @recorded
    def createTransformMatrix_forward(q, s, valid_length):
        return z(3, 3, s.shape[1])

    @recorded
    def createTransformMatrix_backward(g, q, s, valid_length):
        return [z(4, s.shape[1]), z(3, s.shape[1])]

    @recorded
    def jacobianRayspace(view_pos, proj, h, w, valid_length):
        return z(view_pos.shape[0], 3, 3, view_pos.shape[2])

    @recorded
    def createCov2dDirectly_forward(J, view, T, valid_length):
        return z(J.shape[0], 2, 2, J.shape[3])

    @recorded
    def createCov2dDirectly_backward(g, J, view, T, valid_length):
        return z(3, 3, J.shape[3])

    @recorded
    def eigh_and_inv_2x2matrix_forward(x, valid_length):
        V, N = x.shape[0], x.shape[3]
        return [z(V, 2, N), z(V, 2, 2, N), z(V, 2, 2, N)]

    @recorded
    def inv_2x2matrix_backward(inv, g, valid_length):
        return torch.zeros_like(g)

    @recorded
    def get_allocate_size(ndc, view_z, inv_cov, opacity, h, w, th, tw, valid_length):
        V, N = ndc.shape[0], ndc.shape[2]
        return [z(V, 2, N, dtype=i32), z(V, 2, N, dtype=i32), torch.ones((V, N), dtype=i32)]

    @recorded
    def create_table(ndc, inv_cov, opacity, prefix, sorted_id, feedback, idx, h, w, th, tw):
        V = ndc.shape[0]
        L = int(prefix[:, -1].max())
        return [torch.ones((V, L), dtype=i32), z(V, L, dtype=i32)]

    @recorded
    def tileRange(table, max_tile):
        return torch.full((table.shape[0], int(max_tile) + 2), -1, dtype=i32)

    @recorded
    def rasterize_forward(sorted_points, start_index, ndc, cov2d_inv, color, opacity, tiles, h, w, th, tw, enable_stat, enable_trans, enable_depth):
        V, N = ndc.shape[0], ndc.shape[2]
        Hp, Wp, _ = tiles_of(h, w, th, tw)
        return [z(V, 3, Hp, Wp), z(V, 1, Hp, Wp), z(V, 1, Hp, Wp) if enable_depth else torch.zeros((0, 0, 0, 0)), z(V, 1, Hp, Wp, dtype=torch.int16),
                z(V, N, 16), z(V, 1, N, dtype=i32), z(V, 1, N)]

    @recorded
    def rasterize_backward(sorted_points, start_index, packed, tiles, final_t, last, d_img, d_trans, d_depth, inv_scaler, h, w, th, tw, enable_stat):
        V, N = packed.shape[0], packed.shape[1]
        return [z(V, 4, N), z(V, 2, 2, N), z(V, 3, N), z(1, N), z(V, 1, N), z(V, 1, N)]

    @recorded
    def adamUpdate(param, grad, exp_avg, exp_avg_sq, visible_index, valid_length, lr, b1, b2, eps):
        return None

    for f in (frustum_culling_aabb, cull_compact_activate, activate_backward, mvp_transform_forward, mvp_transform_backward,
              createTransformMatrix_forward, createTransformMatrix_backward, jacobianRayspace, createCov2dDirectly_forward,
              createCov2dDirectly_backward, eigh_and_inv_2x2matrix_forward, inv_2x2matrix_backward, get_allocate_size, create_table, tileRange,
              rasterize_forward, rasterize_backward, adamUpdate):
        setattr(m, f.__name__, f)
    return m


class CudaIsCpu(torch.overrides.TorchFunctionMode):
    """this container has no GPU: tensors the reference places on 'cuda' (statistic_helper.py:34, wrapper.py:199) are created on the CPU
    instead -- the environment is bent, not the reference's source"""
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and torch.device(dev).type == "cuda":
            kwargs["device"] = "cpu"
        if getattr(func, "__name__", "") == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        return func(*args, **kwargs)


def load_reference(shim):
    """the reference's layers with package skeletons instead of litegs/__init__.py (which pulls in dataset / image I/O packages)"""
    sys.dont_write_bytecode = True
    sys.modules["litegs_fused"] = shim
    pkg = types.ModuleType("litegs")
    pkg.__path__ = [os.path.join(REF, "litegs")]
    sys.modules["litegs"] = pkg
    for missing in ("cv2", "plyfile", "fused_ssim", "simple_knn", "simple_knn._C"):
        try:
            importlib.import_module(missing)
        except ImportError:
            sys.modules[missing] = types.ModuleType(missing)
    for sub in ("training", "scene"):
        mod = types.ModuleType(f"litegs.{sub}")
        mod.__path__ = [os.path.join(REF, "litegs", sub)]
        sys.modules[f"litegs.{sub}"] = mod
    wrapper = importlib.import_module("litegs.utils.wrapper")
    assert wrapper.litegs_fused is shim
    cluster = importlib.import_module("litegs.scene.cluster")
    sys.modules["litegs.scene"].cluster = cluster
    render = importlib.import_module("litegs.render")
    arguments = importlib.import_module("litegs.arguments")
    optimizer = importlib.import_module("litegs.training.optimizer")
    return wrapper, render, arguments, optimizer, cluster


def normalise(trace, consts):
    """data-dependent sizes -> symbols: a dimension that is not one of the iteration's structural constants becomes "*" """
    def nd(d):
        if d is None:
            return None
        if isinstance(d, list):
            return [nd(x) for x in d]
        if "shape" in d:
            return {"t": d["t"], "shape": [s if s in consts else "*" for s in d["shape"]], "contig": d["contig"]}
        return d
    return [{"op": c["op"], "args": nd(c["args"]), "kwargs": {k: nd(v) for k, v in c["kwargs"].items()}, "out": nd(c["out"])} for c in trace]


CASE = dict(chunks=40, S=128, H=200, W=320, degree=3, tile=(8, 16))


def structural_constants(case=CASE):
    th, tw = case["tile"]
    gy, gx = math.ceil(case["H"] / th), math.ceil(case["W"] / tw)
    return sorted({0, 1, 2, 3, 4, 6, 7, 15, 16, 45, case["S"], case["H"], case["W"], gy * th, gx * tw, gx * gy + 2})


def one_iteration(render, arguments, optimizer, cluster, case=CASE):
    """trainer.py:129-158 on a synthetic chunked cloud (CPU tensors; values are irrelevant to the pattern)"""
    torch.manual_seed(0)
    C, S, H, W, degree = case["chunks"], case["S"], case["H"], case["W"], case["degree"]
    pp = arguments.PipelineParams.__new__(arguments.PipelineParams)      # class attributes carry the defaults
    op = arguments.OptimizationParams.__new__(arguments.OptimizationParams)
    P = lambda *s: torch.nn.Parameter(torch.randn(*s))
    xyz, scale, rot, sh_0, sh_rest, opacity = P(3, C, S), P(3, C, S), P(4, C, S), P(1, 3, C, S), P((degree + 1) ** 2 - 1, 3, C, S), P(1, C, S)
    opt, sched = optimizer.get_optimizer(xyz, scale, rot, sh_0, sh_rest, opacity, 1.0, op, pp)
    view = torch.eye(4)[None].contiguous()
    proj = torch.eye(4)[None].contiguous()
    planes = torch.zeros((1, 6, 4))
    with torch.no_grad():
        origin, extend = cluster.get_cluster_AABB(xyz, scale.exp(), torch.nn.functional.normalize(rot, dim=0))
    TRACE.clear()                                      # (the chunk bounding boxes are computed once per epoch, trainer.py:80, not per iteration)
    vis_id, vis_num, cx, cs, cr, cc, co = render.render_preprocess(origin, extend, planes, view, xyz, scale, rot, sh_0, sh_rest, opacity, None, None, pp, degree)
    img, trans, depth, normal, prim_vis = render.render(view, proj, cx, cs, cr, cc, co, vis_num * pp.cluster_size, None, None, degree, (H, W), pp)
    (img * torch.randn_like(img)).sum().backward()
    opt.step(vis_id, vis_num, prim_vis)
    opt.zero_grad(set_to_none=True)
    sched.step()


def main():
    shim = make_shim()
    import torch.cuda.nvtx as nvtx                     # (no CUDA here: the reference's nvtx ranges become no-ops)
    nvtx.range_push = lambda *a, **k: None
    nvtx.range_pop = lambda *a, **k: None
    with CudaIsCpu():
        wrapper, render, arguments, optimizer, cluster = load_reference(shim)
        one_iteration(render, arguments, optimizer, cluster)
    trace = normalise(TRACE, structural_constants())
    meta = {"reference_files": ["litegs/utils/wrapper.py", "litegs/render/__init__.py", "litegs/training/optimizer.py", "litegs/scene/cluster.py"],
            "iteration": "render_preprocess -> render -> backward -> SparseGaussianAdam.step (litegs/training/trainer.py:129-158)",
            "case": {k: (list(v) if isinstance(v, tuple) else v) for k, v in CASE.items()},
            "calls": len(trace), "ops": [c["op"] for c in trace]}
    with open(OUT, "w") as f:
        json.dump({"meta": meta, "trace": trace}, f, indent=1)
        f.write("\n")
    print(f"{len(trace)} litegs_fused calls recorded -> {OUT}")
    for c in trace:
        print("  ", c["op"])


if __name__ == "__main__":
    main()
