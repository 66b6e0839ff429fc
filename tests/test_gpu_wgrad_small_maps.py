"""fp16 weight gradient on feature maps of at most 16 columns (the discriminator's deep layers,
architecture.py:100-118 of the reference): the kernel packs 2 / 4 / 8 images into one 32-column tile.  Checked
against autograd's conv weight gradient on the same fp16-rounded operands, both the deterministic two-stage form
and the atomic form, including batch tails that do not fill a tile and widths that are not powers of two."""
import numpy as np
import pytest
import torch

from tests.conftest import checks  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


CASES = [  # B, cin, cout, ks, stride, Hout, Wout
    (5, 64, 64, 3, 1, 8, 8), (7, 96, 64, 3, 1, 4, 4), (3, 64, 128, 3, 1, 16, 16), (6, 64, 64, 3, 1, 5, 6),
    (32, 64, 64, 3, 1, 8, 8), (9, 64, 64, 3, 1, 13, 12),
    (5, 64, 64, 4, 2, 4, 4), (7, 64, 96, 4, 2, 8, 8), (3, 128, 64, 4, 2, 16, 16), (6, 64, 64, 4, 2, 3, 6),
    (32, 64, 64, 4, 2, 4, 4),
]


@pytest.mark.parametrize('partial', [True, False])
@pytest.mark.parametrize('case', CASES)
def test_packed_wgrad_matches_autograd(dev, case, partial):
    from esrganplus_amd import engine as E, _lib as L
    B, cin, cout, ks, st, Ho, Wo = case
    Hi, Wi = (Ho, Wo) if st == 1 else (2 * Ho, 2 * Wo)
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = torch.from_numpy(rng.standard_normal((B, cin, Hi, Wi), dtype=np.float32)).half().float()
    g = torch.from_numpy(rng.standard_normal((B, cout, Ho, Wo), dtype=np.float32)).half().float()
    w = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    y = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=st, padding=(ks - 1) // 2)
    y.backward(g.double())
    xin = E.G32(B, cin, Hi, Wi, 'fp16', dev)
    gin = E.G32(B, cout, Ho, Wo, 'fp16', dev)
    xd, gd = x.to(dev), g.to(dev)
    dw = torch.zeros(cout, cin, ks, ks, device=dev)
    db = torch.zeros(cout, device=dev)
    ops = L.OpList()
    for t, buf, C_, H, W in ((xd, xin, cin, Hi, Wi), (gd, gin, cout, Ho, Wo)):
        lo = L.esr_layout()
        lo.dtype, lo.to_g32, lo.B, lo.C, lo.H, lo.W, lo.nchw, lo.g32 = L.ESR_F16, 1, B, C_, H, W, t.data_ptr(), buf.view(0, C_)
        ops.add(L.OP_LAYOUT, 'layout', lo)
    wg = L.esr_wgrad()
    wg.dtype, wg.ks, wg.stride, wg.upsample = L.ESR_F16, ks, st, 0
    wg.B, wg.H, wg.W, wg.cout, wg.cin = B, Ho, Wo, cout, cin
    wg.g, wg.in_ = gin.view(0, cout), xin.view(0, cin)
    wg.dw, wg.dbias, wg.scale = dw.data_ptr(), db.data_ptr(), 1.0
    ops.add(L.OP_WGRAD, 'wgrad', wg)
    arena = E.attach_wgrad_arena(ops, dev) if partial else None
    assert (arena is not None) == partial
    ops.run(E.current_stream())
    torch.cuda.synchronize()
    ref_w, ref_b = w.grad, b.grad
    tol = 2e-3 * ref_w.abs().max().item()
    assert (dw.cpu() - ref_w).abs().max().item() <= tol
    assert (db.cpu() - ref_b).abs().max().item() <= 2e-3 * ref_b.abs().max().item()
