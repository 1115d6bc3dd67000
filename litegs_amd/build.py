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
    "raster.hip": ["-munsafe-fp-atomics"],
    "loss.hip": ["-munsafe-fp-atomics"],
    "fused.hip": ["-ffp-contract=off"],
    "knn.hip": [],
    "refine.hip": ["-ffp-contract=off"],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src: str, dst: str) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


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
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
