"""autograd binding of csrc/loss.hip (fused L1 + SSIM loss, forward and backward on the GPU, no host sync)."""
from __future__ import annotations

import torch

from ._lib import check, lib
from .loss import LAMBDA_DSSIM


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


class FusedL1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img: torch.Tensor, gt: torch.Tensor):
        if not (img.is_cuda and gt.is_cuda):
            raise RuntimeError("fused_l1_ssim_loss: GPU tensors required (no CPU path)")
        img = img.contiguous()
        gt = gt.contiguous()
        if img.shape != gt.shape or img.dtype != torch.float32 or gt.dtype != torch.float32:
            raise RuntimeError("fused_l1_ssim_loss: img and gt must be float32 tensors of the same shape [B,C,H,W]")
        B, C, H, W = img.shape
        L = lib()
        dmaps = torch.empty((3, B * C, H, W), dtype=torch.float32, device=img.device)
        partial = torch.empty((L.lg_l1_ssim_partial_floats(B * C, H, W),), dtype=torch.float32, device=img.device)
        loss = torch.empty((), dtype=torch.float32, device=img.device)
        check(L.lg_l1_ssim_forward(img.data_ptr(), gt.data_ptr(), B * C, H, W, LAMBDA_DSSIM, dmaps.data_ptr(), partial.data_ptr(),
                                   loss.data_ptr(), _s()), "l1_ssim_forward")
        ctx.save_for_backward(img, gt, dmaps)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        img, gt, dmaps = ctx.saved_tensors
        B, C, H, W = img.shape
        d_img = torch.empty_like(img)
        g = grad_out.contiguous()
        check(lib().lg_l1_ssim_backward(img.data_ptr(), gt.data_ptr(), dmaps.data_ptr(), g.data_ptr(), B * C, H, W, LAMBDA_DSSIM,
                                        d_img.data_ptr(), _s()), "l1_ssim_backward")
        return d_img, None


class RasterL1SSIM(torch.autograd.Function):
    """loss(clamp(raw[..., :H, :W], 0, 1), gt) taken directly on the executor's raw tile-padded raster image: the clamp, the crop and
    their backward (four elementwise launches + a zero-filled padded gradient in plain torch) live inside the two loss kernels."""

    @staticmethod
    def forward(ctx, raw: torch.Tensor, gt: torch.Tensor, value_in_backward: bool = False):
        """value_in_backward: the loss VALUE is written by the backward kernel (one launch less per training step); until backward()
        has run the returned tensor is uninitialised -- for loops that call backward() right away and read the value afterwards."""
        if not (raw.is_cuda and gt.is_cuda) or raw.dtype != torch.float32 or gt.dtype != torch.float32:
            raise RuntimeError("raster_l1_ssim_loss: float32 GPU tensors required (no CPU path)")
        if not raw.is_contiguous():
            raise RuntimeError("raster_l1_ssim_loss: the raw raster image must be contiguous")
        gt = gt.contiguous()
        B, C, H, W = gt.shape
        Hp, Wp = raw.shape[-2], raw.shape[-1]
        if raw.shape[:2] != gt.shape[:2] or Hp < H or Wp < W:
            raise RuntimeError("raster_l1_ssim_loss: raw must be [B,C,Hp>=H,Wp>=W]")
        L = lib()
        dmaps = torch.empty((3, B * C, H, W), dtype=torch.float32, device=raw.device)
        partial = torch.empty((L.lg_l1_ssim_partial_floats(B * C, H, W),), dtype=torch.float32, device=raw.device)
        loss = torch.empty((), dtype=torch.float32, device=raw.device)
        check(L.lg_l1_ssim_forward_raster(raw.data_ptr(), Hp, Wp, gt.data_ptr(), B * C, H, W, LAMBDA_DSSIM, dmaps.data_ptr(),
                                          partial.data_ptr(), None if value_in_backward else loss.data_ptr(), _s()), "l1_ssim_forward_raster")
        ctx.save_for_backward(raw, gt, dmaps)
        ctx.pending = (partial, loss) if value_in_backward else None
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        raw, gt, dmaps = ctx.saved_tensors
        B, C, H, W = gt.shape
        Hp, Wp = raw.shape[-2], raw.shape[-1]
        d_raw = torch.empty_like(raw)
        g = grad_out.contiguous()
        partial, loss = ctx.pending if ctx.pending is not None else (None, None)
        ctx.pending = None              # breaks the cycle loss -> grad_fn -> ctx -> loss (one leaked step per iteration otherwise)
        check(lib().lg_l1_ssim_backward_raster_value(raw.data_ptr(), Hp, Wp, gt.data_ptr(), dmaps.data_ptr(), g.data_ptr(), B * C, H, W,
                                                     LAMBDA_DSSIM, d_raw.data_ptr(), partial.data_ptr() if partial is not None else None,
                                                     loss.data_ptr() if loss is not None else None, _s()), "l1_ssim_backward_raster")
        return d_raw, None, None


def raster_l1_ssim_loss(raw: torch.Tensor, gt: torch.Tensor, value_in_backward: bool = False) -> torch.Tensor:
    return RasterL1SSIM.apply(raw, gt, value_in_backward)
