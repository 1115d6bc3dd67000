"""Which binding of the ``litegs_fused`` surface the host mirror (wrapper.py, render.py, optimizer.py, ...) calls.

Two bindings of the same C ABI exist: the compiled torch extension ``_litegs_fused_C`` (csrc/ext/litegs_fused_ext.cpp; what the
reference ships as ``litegs_fused``, GR/ext_cuda.cpp) and the ctypes module ``fused.py``.  ``ops`` exposes the 26 reference names
from the compiled module when it is built (a call costs a few microseconds of host time instead of tens) and everything else
(helpers, the DP extension of ``adamUpdate``) from ``fused.py``.  ``LITEGS_FUSED_BINDING=ctypes`` forces the ctypes binding.
Both bindings are HIP paths; neither is a fallback for the other's results.
"""
from __future__ import annotations

import importlib
import os
import types

from . import fused as _py


def _load_compiled():
    if os.environ.get("LITEGS_FUSED_BINDING", "ext") == "ctypes":
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_litegs_fused_C.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (the extension links against libtorch)
    from ._lib import lib
    lib()                                  # liblitegs_hip.so first: the extension resolves its lg_* symbols from it
    return importlib.import_module("litegs_amd._litegs_fused_C")


compiled = _load_compiled()


def _make_ops():
    ns = types.SimpleNamespace(**{k: getattr(_py, k) for k in dir(_py) if not k.startswith("__")})
    if compiled is not None:
        for name in _py.EXPORTS:
            setattr(ns, name, getattr(compiled, name))
        c_adam = compiled.adamUpdate

        def adamUpdate(param, param_grad, exp_avg, exp_avg_sq, visible_index, valid_length, lr, b1, b2, eps, grad_dense=False):
            if grad_dense:                 # DP extension (dense gradient indexed by chunk id): only the ctypes binding has it
                return _py.adamUpdate(param, param_grad, exp_avg, exp_avg_sq, visible_index, valid_length, lr, b1, b2, eps, grad_dense=True)
            return c_adam(param, param_grad, exp_avg, exp_avg_sq, visible_index, valid_length, lr, b1, b2, eps)
        ns.adamUpdate = adamUpdate
    ns.binding = "ext" if compiled is not None else "ctypes"
    return ns


ops = _make_ops()
