"""HIP fused L1+SSIM loss vs the plain torch fp32 statement of the same formula (the reference's loss lives in an
un-vendored submodule: parity unpinned, formula stated in litegs_amd/loss.py)."""
import pytest
import torch

from tests.torch_loss import l1_ssim_loss_torch

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
    l_ref = l1_ssim_loss_torch(img.double(), gt.double())
    (l_ref * 3.0).backward()
    g_ref = img.grad
    assert abs(l_hip.item() - l_ref.item()) < 2e-6 * max(1.0, abs(l_ref.item()))
    scale = g_ref.abs().max().item()
    assert (g_hip - g_ref).abs().max().item() < 2e-5 * scale


@pytest.mark.parametrize("H,W,Hp,Wp", [(70, 100, 72, 112), (64, 64, 64, 64), (33, 47, 40, 48)])
def test_raster_loss_equals_clamp_crop_then_loss(H, W, Hp, Wp):
    """raster_l1_ssim_loss(raw, gt) == loss(clamp(raw[..., :H, :W], 0, 1), gt), values and gradient w.r.t. the raw padded image
    (zero in the padding and where the clamp saturates)."""
    from litegs_amd import loss as Lm, loss_hip
    g = torch.Generator().manual_seed(3)
    raw = (torch.rand((1, 3, Hp, Wp), generator=g) * 1.6 - 0.3).cuda().requires_grad_(True)      # ~20% below 0, ~20% above 1
    gt = torch.rand((1, 3, H, W), generator=g).cuda()
    l_hip = loss_hip.raster_l1_ssim_loss(raw, gt)
    l_hip.backward()
    g_hip = raw.grad.clone()
    raw64 = raw.detach().double().requires_grad_(True)
    l_ref = l1_ssim_loss_torch(raw64[..., :H, :W].clamp(0, 1), gt.double())
    l_ref.backward()
    assert abs(l_hip.item() - l_ref.item()) < 1e-5
    assert (g_hip.double() - raw64.grad).abs().max().item() < 1e-4 * raw64.grad.abs().max().item() + 1e-9
    assert g_hip[..., H:, :].abs().max().item() == 0 if Hp > H else True
    outside = (raw.detach() < 0) | (raw.detach() > 1)
    assert g_hip[outside].abs().max().item() == 0


def test_loss_value_written_by_the_backward_is_the_same_value():
    """the training step's pairing (forward without the reduction launch, the backward also sums the partial sums): bit-identical loss value
    and gradient"""
    from litegs_amd import loss_hip
    torch.manual_seed(3)
    H, W, Hp, Wp = 270, 480, 272, 480
    raw = (torch.rand((1, 3, Hp, Wp), device="cuda") * 1.4 - 0.2).requires_grad_(True)
    gt = torch.rand((1, 3, H, W), device="cuda")
    one = torch.ones((), device="cuda")
    a = loss_hip.raster_l1_ssim_loss(raw, gt)
    a.backward(one)
    ga = raw.grad.clone(); raw.grad = None
    b = loss_hip.raster_l1_ssim_loss(raw, gt, value_in_backward=True)
    b.backward(one)
    torch.cuda.synchronize()
    assert torch.equal(a.detach(), b.detach()) and torch.equal(ga, raw.grad)
