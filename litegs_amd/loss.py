"""Training loss ``fused_ssim.fused_l1_ssim_loss(img, gt)`` (litegs/training/trainer.py:145).

The reference gets this from an un-vendored submodule (kemchenj/fused-ssim, branch fused-l1-ssim-loss,
``.gitmodules:1-4``; the directory is empty in the reference tree), so its arithmetic is NOT pinned by
anything in /root/reference: "parity unpinned".  Assumed formula (stated in DESIGN.md):
    loss = (1 - lambda) * mean|img - gt| + lambda * (1 - mean SSIM(img, gt)),   lambda = 0.2,
SSIM with an 11x11 Gaussian window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2, zero "same" padding, per channel.
``fused_l1_ssim_loss`` runs the hand-written HIP kernels of csrc/loss.hip (no torch fallback in the product); the plain-torch
statement of the same formula that checks them lives with the tests (tests/torch_loss.py).
"""
from __future__ import annotations

import torch

LAMBDA_DSSIM = 0.2


def fused_l1_ssim_loss(img: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    from . import loss_hip
    return loss_hip.FusedL1SSIM.apply(img, gt)
