"""ctypes loader for ``liblitegs_hip.so`` (the C-ABI HIP library declared in ``include/litegs_hip.h``).

The prototypes are parsed from the header, so the header is the single source of truth for the ABI and
``tests/test_abi.py`` can check that every declared symbol is exported.  There is NO fallback: if the
library is missing or a launch fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LITEGS_HIP_LIB") or os.path.join(_HERE, "liblitegs_hip.so")      # (override: A/B builds of the SAME library, tools/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "litegs_hip.h")

_SCALARS = {
    "int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "double": ctypes.c_double,
    "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32, "uint8_t": ctypes.c_uint8,
    "void": None,
}


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every ``lg_*`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|long long|void)\s*(\*?)\s*\b(lg_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, ptr, name, args = m.group(1), m.group(2), m.group(3), m.group(4).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                a = re.sub(r"\bconst\b", "", a).strip()
                tname = " ".join(a.split()[:-1]) if len(a.split()) > 1 else a
                if tname not in _SCALARS:
                    raise ValueError(f"cannot parse argument '{a}' of {name}")
                argtypes.append(_SCALARS[tname])
        protos[name] = (ctypes.c_void_p if ptr else _SCALARS[ret], argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m litegs_amd.build` (hipcc, gfx950). "
                "litegs_amd has no CPU fallback.")
        # torch first: its wheel bundles the HIP runtime (libamdhip64) it was built against; loading this library before
        # torch would bind the system copy and leave two runtimes in one process ("no ROCm-capable device" at first launch).
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)           # AttributeError if the header declares a missing symbol
            fn.restype = ret
            fn.argtypes = argtypes
            setattr(self, name, fn)


_instance = None


def lib() -> _Lib:
    global _instance
    if _instance is None:
        _instance = _Lib()
    return _instance


_hip = None


def hip_error_string(code: int) -> str:
    global _hip
    try:
        if _hip is None:
            _hip = ctypes.CDLL("libamdhip64.so")
            _hip.hipGetErrorString.restype = ctypes.c_char_p
            _hip.hipGetErrorString.argtypes = [ctypes.c_int]
        return _hip.hipGetErrorString(code).decode()
    except Exception:  # pragma: no cover
        return "hip error"


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"litegs_amd: {what} failed with hipError {code} ({hip_error_string(code)})")
