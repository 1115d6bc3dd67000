"""HIP fused L1+SSIM loss vs the plain torch fp32 statement of the same formula (the reference's loss lives in an
un-vendored submodule: parity unpinned, formula stated in litegs_amd/loss.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (1, 3, 141, 250), (2, 3, 33, 17), (1, 3, 1080, 1920)])
def test_l1_ssim_loss_matches_torch(shape):
    from litegs_amd import loss as Lm
    g = torch.Generator().manual_seed(0)
    img = torch.rand(shape, generator=g).cuda().requires_grad_(True)
    gt = torch.rand(shape, generator=g).cuda()
    gt[..., : shape[2] // 2, :] = img.detach()[..., : shape[2] // 2, :] * 0.9 + 0.05      # correlated half: SSIM far from 0
    l_hip = Lm.fused_l1_ssim_loss(img, gt)
    (l_hip * 3.0).backward()
    g_hip = img.grad.clone()
    img.grad = None
    l_ref = Lm.l1_ssim_loss_torch(img.double(), gt.double())
    (l_ref * 3.0).backward()
    g_ref = img.grad
    assert abs(l_hip.item() - l_ref.item()) < 2e-6 * max(1.0, abs(l_ref.item()))
    scale = g_ref.abs().max().item()
    assert (g_hip - g_ref).abs().max().item() < 2e-5 * scale
