"""The reference's call pattern at the `litegs_fused` boundary, replayed on the GPU through this repository's mirror of the reference's layers.

tests/golden/reference_call_trace.json is a recording of the UNMODIFIED reference Python (litegs/utils/wrapper.py, litegs/render/__init__.py,
litegs/training/optimizer.py, imported from /root/reference by tests/golden/make_call_trace.py in the build container) driving one training
iteration: which of the 26 operators it calls, in which order, how often, and with which dtypes / ranks / layouts -- int64 depth-sorted ids from
torch.sort, an int32 cumsum as offsets, six separate adamUpdate calls.  The Python reference cannot travel to the GPU box; its pattern can.
Here the same iteration runs through litegs_amd.render / wrapper / optimizer with `LITEGS_OPERATOR_BINNING=reference` (litegs_amd/binning.py
reference_pattern_table) on the real kernels, every boundary call is recorded the same way, and the two sequences must be identical.  That mode is
what bench.py times as `reference_call_pattern_ms`; its image and gradients are checked against the oracle below."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from tests.util import case, compacted_grads, oracle_forward, parity_image_and_gradients

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _trace_tools():
    spec = importlib.util.spec_from_file_location("make_call_trace", os.path.join(HERE, "golden", "make_call_trace.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)            # (defines functions only; the reference is imported by its main())
    return mod


def _fmt(d):
    if d is None:
        return "None"
    if isinstance(d, list):
        return "[" + ", ".join(_fmt(x) for x in d) + "]"
    if "shape" in d:
        return f"{d['t']}{d['shape']}" + ("" if d["contig"] else "(strided)")
    return str(d)


def test_mirror_drives_the_boundary_exactly_as_the_reference_does(oracle):
    from litegs_amd import binning as B, optimizer as OPT, render as R
    from litegs_amd.binding import ops
    from litegs_amd import fused as PY
    tools = _trace_tools()
    fixture = json.load(open(os.path.join(HERE, "golden", "reference_call_trace.json")))
    want = fixture["trace"]
    c = case("small")
    H, W = c["H"], c["W"]
    assert (H, W) == (fixture["meta"]["case"]["H"], fixture["meta"]["case"]["W"]) and c["degree"] == fixture["meta"]["case"]["degree"]
    res = oracle_forward("small")
    params = [torch.nn.Parameter(torch.from_numpy(p).cuda()) for p in c["params"]]
    view, proj, planes = [torch.from_numpy(x).cuda() for x in (c["view"], c["proj"], c["planes"])]
    pp = R.PipelineParams()
    opt, sched = OPT.get_optimizer(*[params[i] for i in (0, 1, 2, 3, 4, 5)], 1.0, OPT.OptimizationParams())
    with torch.no_grad():
        origin, extend = R.get_cluster_AABB(params[0], params[1].exp(), torch.nn.functional.normalize(params[2], dim=0))
    rng = np.random.default_rng(4)
    w_host = rng.standard_normal((1, 3, H, W)).astype(np.float32)

    tools.TRACE.clear()
    saved, mode = {}, B._MODE
    for name in PY.EXPORTS:
        fn = getattr(ops, name)
        saved[name] = fn
        def make(nm, f):
            def call(*a, **k):
                out = f(*a, **k)
                tools.TRACE.append({"op": nm, "args": [tools._desc(x) for x in a], "kwargs": {kk: tools._desc(v) for kk, v in k.items()}, "out": tools._desc(out)})
                return out
            return call
        setattr(ops, name, make(name, fn))
    B._MODE, grouped = "reference", B._GROUPED
    B._GROUPED = False
    try:
        vis_id, vis_num, cx, cs, cr, cc, co = R.render_preprocess(origin, extend, planes, view, *params, None, None, pp, c["degree"])
        img, trans, depth, normal, prim_vis = R.render(view, proj, cx, cs, cr, cc, co, vis_num * pp.cluster_size, None, None, c["degree"], (H, W), pp)
        (img * torch.from_numpy(w_host).cuda()).sum().backward()
        torch.cuda.synchronize()
        grads = [p.grad for p in params]
        like = oracle.render_backward(res, c["params"], c["view"], c["proj"], np.zeros_like(res.img), H, W, c["degree"])[0]
        got_grads = compacted_grads(params, res.nvis, like)
        got_img = img.detach().cpu().numpy()
        opt.step(vis_id, vis_num, prim_vis)
        opt.zero_grad(set_to_none=True)
        sched.step()
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
        B._MODE, B._GROUPED = mode, grouped
    got = tools.normalise(list(tools.TRACE), tools.structural_constants())
    tools.TRACE.clear()
    assert grads is not None

    lines = []
    for i in range(max(len(got), len(want))):
        g = got[i] if i < len(got) else None
        r = want[i] if i < len(want) else None
        same = g is not None and r is not None and g["op"] == r["op"] and g["args"] == r["args"] and g["kwargs"] == r["kwargs"] and g["out"] == r["out"]
        if not same:
            lines.append(f"call {i}:\n    reference: {r and r['op']}({', '.join(_fmt(a) for a in (r['args'] if r else []))}) -> {_fmt(r['out']) if r else ''}\n"
                         f"    mirror   : {g and g['op']}({', '.join(_fmt(a) for a in (g['args'] if g else []))}) -> {_fmt(g['out']) if g else ''}")
    assert not lines, "the mirror's boundary calls differ from the reference's recorded pattern:\n" + "\n".join(lines)
    assert [c_["op"] for c_ in got] == fixture["meta"]["ops"]
    # and what those calls computed: image and all six gradients against the oracle (torch.sort is not stable: equal depths may be ordered
    # differently from the oracle's stable order -- there are none in this cloud)
    parity_image_and_gradients(oracle, res, got_img, got_grads, c["params"], c["view"], c["proj"], w_host, H, W, c["degree"], tag=" (reference pattern)")
