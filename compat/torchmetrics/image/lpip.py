import warnings

import torch


class LearnedPerceptualImagePatchSimilarity(torch.nn.Module):
    """LPIPS needs pretrained network weights (the real package downloads them); there is no network here.  Returns NaN (one warning)
    so that the reference's metrics script still runs to its end and prints PSNR / SSIM."""
    _warned = False

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, preds, target):
        if not LearnedPerceptualImagePatchSimilarity._warned:
            warnings.warn("compat/torchmetrics: LPIPS is not available (needs pretrained weights); reporting NaN")
            LearnedPerceptualImagePatchSimilarity._warned = True
        return torch.full((), float("nan"), device=preds.device)
