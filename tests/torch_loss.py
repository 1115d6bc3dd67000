"""Plain-torch statement of the training loss (test infrastructure: the checker of csrc/loss.hip, and the loss of the dense torch
reference renderer in tests/convergence.py).  Formula: litegs_amd/loss.py header."""
import torch
import torch.nn.functional as Fn

LAMBDA_DSSIM = 0.2


def _window(device, dtype):
    x = torch.arange(11, device=device, dtype=dtype) - 5
    g = torch.exp(-(x * x) / (2 * 1.5 * 1.5))
    return g / g.sum()


def ssim_map_torch(img: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    C = img.shape[1]
    g = _window(img.device, img.dtype)
    w = (g[:, None] * g[None, :]).expand(C, 1, 11, 11).contiguous()

    def blur(x):
        return Fn.conv2d(x, w, padding=5, groups=C)

    mu1, mu2 = blur(img), blur(gt)
    s11 = blur(img * img) - mu1 * mu1
    s22 = blur(gt * gt) - mu2 * mu2
    s12 = blur(img * gt) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))


def l1_ssim_loss_torch(img: torch.Tensor, gt: torch.Tensor, lam: float = LAMBDA_DSSIM) -> torch.Tensor:
    return (1 - lam) * (img - gt).abs().mean() + lam * (1 - ssim_map_torch(img, gt).mean())
