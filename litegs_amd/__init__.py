"""litegs_amd -- MI355X (gfx950) native implementation of the LiteGS render hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels behind a C ABI, ``include/litegs_hip.h``),
``fused`` (the ``litegs_fused`` operator surface of the reference, GR/ext_cuda.cpp:9-35),
``wrapper``/``render`` (host-side mirror of litegs/utils/wrapper.py and litegs/render/__init__.py),
``optimizer`` (sparse Adam), ``dp`` (one-frame-per-GPU data parallelism over RCCL) and
``synthetic`` (seeded workloads).  There is no CPU fallback: anything that computes raises if the
HIP library is missing.
"""
__version__ = "0.1.0"
