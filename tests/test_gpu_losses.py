"""Fused train-step losses (csrc/loss_kernels.hip, esrganplus_amd/losses.py) against the torch formulas the
reference uses: nn.L1Loss (cri_pix / cri_fea) and GANLoss('vanilla') on the relativistic-average logits
(codes/models/modules/loss.py:6-38, SRRaGAN_model.py:124-137,150-156) — values and gradients."""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import checks  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('shape', [(16, 3, 128, 128), (16, 512, 8, 8), (3, 5, 7, 11), (1, 1, 1, 3)])
def test_l1_loss_and_gradient(dev, shape):
    from esrganplus_amd import losses as LS
    g = torch.Generator().manual_seed(sum(shape))
    a0 = torch.randn(shape, generator=g).to(dev)
    b = torch.randn(shape, generator=g).to(dev)
    b.view(-1)[::7] = a0.view(-1)[::7]                  # exact ties: sign(0) = 0 like torch
    res = []
    for fn in (lambda a: 0.37 * F.l1_loss(a, b), lambda a: LS.l1_loss(a, b, 0.37)):
        a = a0.clone().requires_grad_(True)
        loss = fn(a)
        (loss * 1024.0).backward()
        res.append((loss.detach(), a.grad))
    (l0, g0), (l1, g1) = res
    assert abs(l0.item() - l1.item()) <= 1e-6 * max(1.0, abs(l0.item()))
    assert torch.allclose(g0, g1, rtol=1e-6, atol=0)
    for _ in range(2):                                  # the scratch is left clean: same answer again
        assert LS.l1_loss(a0, b, 0.37).item() == l1.item()


@pytest.mark.parametrize('n', [16, 1, 37, 300])
@pytest.mark.parametrize('mode', ['g_step', 'd_step'])
def test_ragan_loss_and_gradients(dev, n, mode):
    from esrganplus_amd import losses as LS
    g = torch.Generator().manual_seed(n)
    x0 = (3 * torch.randn(n, 1, generator=g)).to(dev)
    y0 = (3 * torch.randn(n, 1, generator=g)).to(dev)
    w = 5e-3 if mode == 'g_step' else 1.0
    tx, ty = (False, True) if mode == 'g_step' else (True, False)

    def ref(x, y):
        t = lambda v, r: torch.ones_like(v) if r else torch.zeros_like(v)
        lx = F.binary_cross_entropy_with_logits(x - y.mean(), t(x, tx))
        ly = F.binary_cross_entropy_with_logits(y - x.mean(), t(y, ty))
        return w * (lx + ly) / 2, torch.stack([x.detach().mean(), y.detach().mean(), lx.detach(), ly.detach()])

    out = []
    for fn in (ref, lambda x, y: LS.ragan_loss(x, y, tx, ty, w)):
        x = x0.clone().requires_grad_(mode == 'd_step')           # G step: x = pred_d_real, detached
        y = y0.clone().requires_grad_(True)
        loss, aux = fn(x, y)
        (loss * 1024.0).backward()
        out.append((loss.detach(), aux, x.grad, y.grad))
    (l0, a0, gx0, gy0), (l1, a1, gx1, gy1) = out
    assert abs(l0.item() - l1.item()) <= 2e-6 * max(1e-3, abs(l0.item()))
    assert torch.allclose(a0, a1, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gy0, gy1, rtol=1e-4, atol=1e-7 * 1024 * w)
    if mode == 'd_step':
        assert torch.allclose(gx0, gx1, rtol=1e-4, atol=1e-7 * 1024 * w)
    else:
        assert gx1 is None and not a1.requires_grad
