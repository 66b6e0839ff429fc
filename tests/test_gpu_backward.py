"""GPU parity of the RRDBNet backward (dgrad through the fused conv kernel with transposed weights,
wgrad kernel, LeakyReLU/noise/residual backward in epilogues) against the golden gradients captured
from the imported reference (oracle/gen_golden.py)."""
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


def zs(seed, shapes, tag):
    return [synth.normal_like(seed, '%s.%d' % (tag, i), s) for i, s in enumerate(shapes)]


FULL = ('model.0.weight', 'model.1.sub.0.RDB2.conv2.0.weight', 'model.1.sub.0.RDB3.conv1x1.weight',
        'model.1.sub.0.RDB1.conv5.0.bias', 'model.6.weight', 'model.10.weight', 'model.10.bias')


@pytest.mark.parametrize('tag,nb,shape,variant', [('a', 1, (1, 3, 16, 20), 'codes'),
                                                  ('b', 2, (2, 3, 24, 24), 'codes'),
                                                  ('c', 1, (1, 3, 13, 18), 'test_image')])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_rrdbnet_param_grads_fp32(dev, golden, tag, nb, shape, variant, mode):
    from esrganplus_amd import architecture as arch
    from oracle import ref_torch as RT
    g = golden('rrdbnet_small')
    sd = synth.rrdbnet_state_dict(nb=nb, seed=20 + nb)
    cls = arch.RRDBNet if variant == 'codes' else arch.RRDB_Net
    net = cls(3, 3, 64, nb).to(dev)
    net.load_state_dict(sd, strict=True)
    net.train(mode == 'train')
    x = synth.image_batch(3, *shape, name='small.x.' + tag).to(dev)
    gy = synth.normal_like(3, 'small.gy.' + tag, (shape[0], 3, shape[2] * 4, shape[3] * 4)).to(dev)
    z = None
    if mode == 'train':
        z = [t.to(dev) for t in zs(7, RT.noise_shapes(shape, nb, variant), 'small.z.' + tag)]
    y = net(x, z=z)
    assert np.abs(y.detach().cpu().numpy() - g['%s_y_%s' % (tag, mode)]).max() <= 1e-4
    (y * gy).sum().backward()
    params = dict(net.named_parameters())
    if mode == 'train':
        for k in FULL:
            ref = g['%s_g_%s' % (tag, k)]
            got = params[k].grad.cpu().numpy()
            err = np.abs(got - ref).max()
            assert err <= 2e-3 * max(1.0, np.abs(ref).max()), (k, err, np.abs(ref).max())
    chk = np.stack([checks(params[k].grad) for k in sd.keys()])
    ref = g['%s_gchk_%s' % (tag, mode)]
    rel = np.abs(chk - ref) / np.maximum(1.0, np.abs(ref[:, 1:2]))
    assert rel.max() <= 2e-3, (np.unravel_index(rel.argmax(), rel.shape), rel.max())


def test_backward_fp16_close_to_fp32(dev):
    """fp16 storage path: gradients track the fp32 path (loss-scaled by the caller in training)."""
    from esrganplus_amd import architecture as arch
    sd = synth.rrdbnet_state_dict(nb=1, seed=3)
    x = synth.image_batch(3, 2, 3, 16, 16, name='bw16.x').to(dev)
    gy = synth.normal_like(3, 'bw16.gy', (2, 3, 64, 64)).to(dev)
    grads = {}
    for prec in ('fp32', 'fp16'):
        net = arch.RRDBNet(3, 3, 64, 1).to(dev).eval().set_precision(prec)
        net.load_state_dict(sd)
        (net(x) * gy).sum().backward()
        grads[prec] = {k: p.grad.clone() for k, p in net.named_parameters()}
    for k in grads['fp32']:
        a, b = grads['fp32'][k], grads['fp16'][k]
        rel = ((a - b).norm() / a.norm().clamp_min(1e-6)).item()   # fp16 storage: ~1e-2 relative
        assert rel <= 6e-2, (k, rel)


def test_two_forwards_before_backward_use_distinct_plans(dev):
    from esrganplus_amd import architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1).to(dev).eval()
    net.load_state_dict(synth.rrdbnet_state_dict(1, 4))
    x1 = synth.image_batch(1, 1, 3, 8, 8, name='p.x1').to(dev)
    x2 = synth.image_batch(2, 1, 3, 8, 8, name='p.x2').to(dev)
    y1, y2 = net(x1), net(x2)
    (y1.sum() + 2 * y2.sum()).backward()
    g12 = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad()
    net(x1).sum().backward()
    (2 * net(x2).sum()).backward()
    for k, p in net.named_parameters():
        assert (p.grad - g12[k]).abs().max().item() <= 1e-4 * max(1.0, g12[k].abs().max().item()), k


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_standalone_rdb_backward_golden(dev, golden, mode):
    """ResidualDenseBlock_5C used on its own (block.py:232-268): input gradient and parameter
    gradients against the reference's (tests/golden/rdb.npz, captured from the imported reference)."""
    from esrganplus_amd import block as B
    g = golden('rdb')
    sd = synth.rrdbnet_state_dict(nb=1, seed=11)
    p = 'model.1.sub.0.RDB1.'
    m = B.ResidualDenseBlock_5C(64).to(dev).set_precision('fp32')
    m.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)})
    m.train(mode == 'train')
    x = synth.normal_like(11, 'rdb.x', (1, 64, 12, 12)).to(dev).requires_grad_(True)
    gy = synth.normal_like(11, 'rdb.gy', (1, 64, 12, 12)).to(dev)
    z = zs(5, [(1, 64, 12, 12)], 'rdb.z')[0].to(dev) if mode == 'train' else None
    y = m(x, z=z) if z is not None else m(x)
    assert np.abs(y.detach().cpu().numpy() - g['y_' + mode]).max() <= 2e-5
    (y * gy).sum().backward()
    gx = x.grad.cpu().numpy()
    ref = g['gx_' + mode]
    assert np.abs(gx - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    if mode == 'train':
        for name, par in (('gw_conv1', m.conv1[0].weight), ('gw_conv3', m.conv3[0].weight),
                          ('gw_conv5', m.conv5[0].weight), ('gb_conv4', m.conv4[0].bias),
                          ('gw_conv1x1', m.conv1x1.weight)):
            got, want = par.grad.cpu().numpy(), g[name]
            assert np.abs(got - want).max() <= 2e-3 * max(1e-3, np.abs(want).max()), name


@pytest.mark.parametrize('variant', ['codes', 'test_image'])
def test_standalone_rrdb_backward_vs_oracle(dev, variant):
    """RRDB on its own (block.py:271-291 / test_image/block.py:236-256), train mode with explicit
    noise: input + parameter gradients against the CPU restatement of the reference."""
    from esrganplus_amd import block as B
    from oracle import ref_torch as RT
    sd = synth.rrdbnet_state_dict(nb=1, seed=31)
    p = 'model.1.sub.0.'
    m = B.RRDB(64, extra_noise=(variant == 'test_image')).to(dev).set_precision('fp32').train()
    m.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)})
    shape = (2, 64, 10, 14)
    x = synth.normal_like(31, 'rrdb.x', shape)
    gy = synth.normal_like(31, 'rrdb.gy', shape)
    nz = 4 if variant == 'test_image' else 3
    z = zs(6, [shape] * nz, 'rrdb.z')
    # oracle
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(p)}
    xo = x.clone().requires_grad_(True)
    yo = RT.rrdb_forward(xo, sdr, p[:-1], z[:3], z[3] if nz == 4 else None)
    (yo * gy).sum().backward()
    # HIP
    xg = x.to(dev).requires_grad_(True)
    y = m(xg, z=[t.to(dev) for t in z])
    assert (y.detach().cpu() - yo.detach()).abs().max().item() <= 5e-5
    (y * gy.to(dev)).sum().backward()
    ref = xo.grad
    assert (xg.grad.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    for k in ('RDB1.conv1.0.weight', 'RDB2.conv1x1.weight', 'RDB3.conv5.0.weight', 'RDB2.conv4.0.bias'):
        got = dict(m.named_parameters())[k].grad.cpu()
        want = sdr[p + k].grad
        assert (got - want).abs().max().item() <= 2e-3 * max(1e-3, want.abs().max().item()), k
