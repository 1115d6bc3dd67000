"""Shim for the reference's un-vendored `fused_ssim` submodule (litegs/training/trainer.py:3,145): exposes
``fused_l1_ssim_loss(img, gt)`` backed by the HIP kernels of litegs_amd/csrc/loss.hip.  Formula and its "parity
unpinned" status: litegs_amd/loss.py."""
from litegs_amd.loss import fused_l1_ssim_loss  # noqa: F401
