"""SRResNet x4 and pixelshuffle_block (codes/models/modules/architecture.py:13-44, block.py:299-312; built by
networks.py:88-91 with relu / pixelshuffle, train_SRResNet.json: norm null, CNA) on the per-conv HIP modules
(``Conv2dHIP``): forward, input gradient and parameter gradients against fixtures from the IMPORTED reference
(oracle/gen_golden.py gen_srresnet), both upsamplers."""
import numpy as np
import pytest
import torch

from esrganplus_amd import synth
from tests.conftest import checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('mode', ['pixelshuffle', 'upconv'])
def test_srresnet_golden(dev, golden, mode):
    from esrganplus_amd import architecture as arch
    g = golden('srresnet_' + mode)
    nb = 3
    net = arch.SRResNet(3, 3, 64, nb, upscale=4, norm_type=None, act_type='relu', mode='CNA',
                        upsample_mode=mode).to(dev)
    net.load_state_dict(synth.srresnet_state_dict(nb=nb, seed=77, upsample_mode=mode), strict=True)
    x = synth.image_batch(77, 2, 3, 20, 28, name='srresnet.x').to(dev)
    gy = synth.normal_like(77, 'srresnet.gy', (2, 3, 80, 112)).to(dev)
    xr = x.clone().requires_grad_(True)
    y = net(xr)
    (y * gy).sum().backward()
    assert np.abs(y.detach().cpu().numpy() - g['y']).max() <= 1e-4
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert rel(xr.grad.cpu().numpy(), g['gx']) <= 2e-3
    params = dict(net.named_parameters())
    for key in ('model.0.weight', 'model.1.sub.1.res.2.bias', 'model.10.weight'):
        assert rel(params[key].grad.cpu().numpy(), g['g_' + key]) <= 2e-3, key
    up = 'model.2.weight' if mode == 'pixelshuffle' else 'model.3.weight'
    assert rel(params[up].grad.cpu().numpy(), g['g_up']) <= 2e-3
    chk = np.stack([checks(params[k].grad) for k in params])
    assert np.abs(chk[:, 2] - g['gchk'][:, 2]).max() <= 2e-3 * g['gchk'][:, 2].max()   # every gradient's L2 norm
    # fp16 storage / fp32 accumulation: same net, PSNR-style gate on the output
    net16 = arch.SRResNet(3, 3, 64, nb, upsample_mode=mode).to(dev).set_precision('fp16')
    net16.load_state_dict(net.state_dict())
    with torch.no_grad():
        y16 = net16(x)
    assert (y16 - y.detach()).abs().max().item() <= 3e-2 * max(1.0, y.detach().abs().max().item())


def test_pixelshuffle_block_matches_torch(dev):
    """block.pixelshuffle_block: conv to 4x the channels, nn.PixelShuffle(2), ReLU — forward and gradients against
    torch's own conv on the same parameters."""
    from esrganplus_amd import block as B
    blk = B.pixelshuffle_block(64, 32, act_type='relu').to(dev)
    assert [k for k, _ in blk.named_parameters()] == ['0.weight', '0.bias']
    x = synth.image_batch(5, 2, 64, 12, 20, name='ps.x').to(dev)
    xr = x.clone().requires_grad_(True)
    y = blk(xr)
    gy = synth.normal_like(5, 'ps.gy', tuple(y.shape)).to(dev)
    (y * gy).sum().backward()
    w, b = blk[0].weight.detach(), blk[0].bias.detach()
    xo = x.clone().requires_grad_(True)
    wo, bo = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = torch.relu(torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(xo, wo, bo, padding=1), 2))
    (yo * gy).sum().backward()
    assert y.shape == (2, 32, 24, 40)
    assert (y - yo).abs().max().item() <= 1e-4
    assert (xr.grad - xo.grad).abs().max().item() <= 2e-3 * xo.grad.abs().max().item()
    assert (blk[0].weight.grad - wo.grad).abs().max().item() <= 2e-3 * wo.grad.abs().max().item()
    assert (blk[0].bias.grad - bo.grad).abs().max().item() <= 2e-3 * bo.grad.abs().max().item()
