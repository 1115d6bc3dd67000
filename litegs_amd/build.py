"""Build ``liblitegs_hip.so`` (the C-ABI HIP library) in-tree with hipcc for gfx950.

``python -m litegs_amd.build`` or ``litegs_amd.build.build()``.  hipcc cross-compiles without a GPU.
The built library stays in ``litegs_amd/`` (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblitegs_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-Wno-deprecated-declarations", "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]
# per-file extra flags: exact (contraction-free) arithmetic where integer/index decisions must match the oracle
SOURCES = {
    "transform.hip": ["-ffp-contract=off"],
    "compact.hip": ["-ffp-contract=off"],
    "binning.hip": ["-ffp-contract=off"],
    "tilesort.hip": [],
    "raster.hip": ["-munsafe-fp-atomics"],
    "loss.hip": ["-munsafe-fp-atomics"],
    "fused.hip": ["-ffp-contract=off"],
    "dp.hip": ["-ffp-contract=off"],
    "knn.hip": [],
    "refine.hip": ["-ffp-contract=off"],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


ABI_HEADER = os.path.join(os.path.dirname(HERE), "include", "litegs_hip.h")
EXT_SRC = os.path.join(CSRC, "ext", "litegs_fused_ext.cpp")
EXT_LIB = os.path.join(HERE, "_litegs_fused_C.so")


def _newer(src: str, dst: str) -> bool:
    """the ABI header is a dependency of every object: ctypes prototypes are parsed from it, so a stale object would be called with
    the new argument list"""
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src, ABI_HEADER, os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """``litegs_fused`` as a compiled torch extension (csrc/ext/litegs_fused_ext.cpp): host-only C++ over the C ABI, built with g++
    against the torch headers and linked to liblitegs_hip.so next to it (rpath $ORIGIN).  The reference builds the module of the same
    name with GR/setup.py."""
    if not (force or _newer(EXT_SRC, EXT_LIB) or os.path.getmtime(LIB) > os.path.getmtime(EXT_LIB)):
        return EXT_LIB
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths("cuda") + [sysconfig.get_paths()["include"], "/opt/rocm/include", os.path.join(os.path.dirname(HERE), "include")]
    libdirs = ce.library_paths("cuda")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_litegs_fused_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations",
           EXT_SRC, "-o", EXT_LIB]
    for p in inc:
        cmd += ["-I", p]
    for p in libdirs:
        cmd += ["-L", p, f"-Wl,-rpath,{p}"]
    cmd += ["-L", HERE, "-l:liblitegs_hip.so", "-Wl,-rpath,$ORIGIN", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python"]
    if verbose:
        print(" ".join(cmd), flush=True)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        sys.stderr.write(p.stdout.decode())
        raise RuntimeError("g++ failed on the litegs_fused extension")
    return EXT_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    relink = force or not os.path.exists(LIB)
    procs = []
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj):
            cmd = [hipcc, "-c", src, "-o", obj] + COMMON + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            relink = True
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {name}")
        if verbose and out:
            sys.stderr.write(out.decode())
    if relink:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    try:
        build_extension(force=force, verbose=verbose)
    except (RuntimeError, OSError, ImportError) as e:
        # no g++ / no torch headers: the C-ABI library above is complete, and litegs_amd/fused.py binds the same 26 names through
        # ctypes (litegs_amd/binding.py picks whichever is there)
        import warnings
        warnings.warn(f"litegs_amd.build: the compiled litegs_fused extension was not built ({e}); using the ctypes binding")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
