import torch
import torch.nn.functional as F


class StructuralSimilarityIndexMeasure(torch.nn.Module):
    """mean SSIM, 11x11 Gaussian window sigma 1.5, K1 0.01, K2 0.03 (torchmetrics defaults); valid region only, like torchmetrics
    (which pads by reflection and then crops the border)."""

    def __init__(self, data_range=None, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03, **_unused):
        super().__init__()
        self.range = (data_range[1] - data_range[0]) if isinstance(data_range, (tuple, list)) else (data_range or 1.0)
        self.k, self.sigma, self.k1, self.k2 = kernel_size, sigma, k1, k2

    def forward(self, preds, target):
        C = preds.shape[1]
        x = torch.arange(self.k, dtype=preds.dtype, device=preds.device) - (self.k - 1) / 2
        g = torch.exp(-(x / self.sigma) ** 2 / 2)
        g = (g / g.sum())
        w = (g[:, None] * g[None, :]).expand(C, 1, self.k, self.k).contiguous()
        blur = lambda t: F.conv2d(t, w, groups=C)                     # noqa: E731
        c1, c2 = (self.k1 * self.range) ** 2, (self.k2 * self.range) ** 2
        mx, my = blur(preds), blur(target)
        sxx, syy, sxy = blur(preds * preds) - mx * mx, blur(target * target) - my * my, blur(preds * target) - mx * my
        ssim = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
        return ssim.mean()
