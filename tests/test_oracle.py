"""CPU tests: the oracle (oracle/ref_torch.py, oracle/conv_ref.c) against the golden vectors that
oracle/gen_golden.py captured from the imported reference.  No GPU, no /root/reference."""
import numpy as np
import pytest
import torch

from esrganplus_amd import synth
from oracle import ref_c, ref_torch as RT
from tests.conftest import checks


def zs(seed, shapes, tag):
    return [synth.normal_like(seed, '%s.%d' % (tag, i), s) for i, s in enumerate(shapes)]


def test_rdb_fwd_bwd(golden):
    g = golden('rdb')
    sd = synth.rrdbnet_state_dict(nb=1, seed=11)
    p = 'model.1.sub.0.RDB1'
    x = synth.normal_like(11, 'rdb.x', (1, 64, 12, 12))
    gy = synth.normal_like(11, 'rdb.gy', (1, 64, 12, 12))
    for mode in ('eval', 'train'):
        z = zs(5, [x.shape], 'rdb.z')[0] if mode == 'train' else None
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(p)}
        xo = x.clone().requires_grad_(True)
        y = RT.rdb_forward(xo, sdr, p, z)
        (y * gy).sum().backward()
        assert np.abs(y.detach().numpy() - g['y_' + mode]).max() <= 1e-6
        assert np.abs(xo.grad.numpy() - g['gx_' + mode]).max() <= 1e-5
        if mode == 'train':
            for name, key in (('gw_conv1', '.conv1.0.weight'), ('gw_conv3', '.conv3.0.weight'),
                              ('gw_conv5', '.conv5.0.weight'), ('gb_conv4', '.conv4.0.bias'),
                              ('gw_conv1x1', '.conv1x1.weight')):
                assert np.abs(sdr[p + key].grad.numpy() - g[name]).max() <= 1e-4, name


@pytest.mark.parametrize('tag,nb,shape,variant', [('a', 1, (1, 3, 16, 20), 'codes'),
                                                  ('b', 2, (2, 3, 24, 24), 'codes'),
                                                  ('c', 1, (1, 3, 13, 18), 'test_image')])
def test_rrdbnet_small(golden, tag, nb, shape, variant):
    g = golden('rrdbnet_small')
    sd = synth.rrdbnet_state_dict(nb=nb, seed=20 + nb)
    x = synth.image_batch(3, *shape, name='small.x.' + tag)
    gy = synth.normal_like(3, 'small.gy.' + tag, (shape[0], 3, shape[2] * 4, shape[3] * 4))
    for mode in ('eval', 'train'):
        z = zs(7, RT.noise_shapes(shape, nb, variant), 'small.z.' + tag) if mode == 'train' else None
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        y = RT.rrdbnet_forward(xo, sdr, nb, z, variant)
        (y * gy).sum().backward()
        assert np.abs(y.detach().numpy() - g['%s_y_%s' % (tag, mode)]).max() <= 2e-6
        assert np.abs(xo.grad.numpy() - g['%s_gx_%s' % (tag, mode)]).max() <= 1e-4
        chk = np.stack([checks(sdr[k].grad) for k in sd.keys()])
        ref = g['%s_gchk_%s' % (tag, mode)]
        assert np.abs(chk - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())


def test_rrdbnet_full_32(golden):
    g = golden('rrdbnet_full')
    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    x = synth.image_batch(0, 1, 3, 32, 32, name='full.x32')
    with torch.no_grad():
        y = RT.rrdbnet_forward(x, sd, 23)
    assert np.abs(y.numpy() - g['y32']).max() <= 1e-5


def test_c_restatement_matches_torch_restatement():
    """plain-C primitives (double accumulation) == torch primitives to fp32 round-off."""
    sd = synth.rrdbnet_state_dict(nb=1, seed=21)
    x = synth.image_batch(3, 1, 3, 10, 9, name='c.x')
    z = zs(8, RT.noise_shapes(x.shape, 1, 'test_image'), 'c.z')
    with torch.no_grad():
        yt = RT.rrdbnet_forward(x, sd, 1, z, 'test_image').numpy()
    yc = ref_c.rrdbnet_forward(x.numpy(), sd, 1, [t.numpy() for t in z], 'test_image')
    assert yc.shape == yt.shape
    assert np.abs(yc - yt).max() <= 2e-5


def test_c_primitives_d_and_vgg_ops():
    g = np.random.default_rng(0)
    x = torch.from_numpy(g.standard_normal((2, 5, 12, 10), dtype=np.float32))
    w = torch.from_numpy(g.standard_normal((7, 5, 4, 4), dtype=np.float32))
    b = torch.from_numpy(g.standard_normal(7, dtype=np.float32))
    yt = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1).numpy()
    assert np.abs(ref_c.conv2d(x.numpy(), w.numpy(), b.numpy(), 2) - yt).max() < 1e-4
    assert np.array_equal(ref_c.maxpool2(x.numpy()), torch.nn.functional.max_pool2d(x, 2).numpy())
    gam, bet = g.standard_normal(5, dtype=np.float32), g.standard_normal(5, dtype=np.float32)
    rm, rv = np.zeros(5, np.float32), np.ones(5, np.float32)
    rmt, rvt = torch.zeros(5), torch.ones(5)
    yt = torch.nn.functional.batch_norm(x, rmt, rvt, torch.from_numpy(gam), torch.from_numpy(bet),
                                        True, 0.1, 1e-5).numpy()
    yc = ref_c.batchnorm(x.numpy(), gam, bet, rm, rv, True)
    assert np.abs(yc - yt).max() < 1e-5
    assert np.abs(rv - rvt.numpy()).max() < 1e-6 and np.abs(rm - rmt.numpy()).max() < 1e-6


def test_discriminator(golden):
    g = golden('disc')
    sd = synth.discriminator_state_dict(seed=4)
    x = synth.image_batch(4, 4, 3, 128, 128, name='disc.x')
    gy = synth.normal_like(4, 'disc.gy', (4, 1))
    with torch.no_grad():
        ye = RT.discriminator_forward(x, {k: v.clone() for k, v in sd.items()}, training=False)
    assert np.abs(ye.numpy() - g['y_eval']).max() <= 1e-4
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
               else v.clone()) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    y = RT.discriminator_forward(xo, sdr, training=True)
    (y * gy).sum().backward()
    assert np.abs(y.detach().numpy() - g['y_train']).max() <= 1e-4
    assert np.abs(xo.grad.numpy()[:, :, ::8, ::8] - g['gx_sub8']).max() <= 1e-4
    assert np.abs(sdr['features.27.weight'].grad.numpy() - g['g_features.27.weight']).max() <= 1e-3
    with torch.no_grad():
        for i in range(3):
            RT.discriminator_forward(x * (0.5 + 0.25 * i), sdr, training=True)
    assert np.abs(sdr['features.15.running_var'].numpy() - g['rv_features.15']).max() <= 1e-5
    assert int(sdr['features.3.num_batches_tracked']) == int(g['nbt']) == 4


@pytest.mark.parametrize('size,batch', [(96, 3), (192, 2)])
def test_discriminator_variants(golden, size, batch):
    """Discriminator_VGG_96 / _192 restatement against the imported reference's outputs."""
    g = golden('disc%d' % size)
    sd = synth.discriminator_state_dict(seed=40 + size, size=size)
    x = synth.image_batch(size, batch, 3, size, size, name='disc%d.x' % size)
    gy = synth.normal_like(size, 'disc%d.gy' % size, (batch, 1))
    with torch.no_grad():
        ye = RT.discriminator_forward(x, {k: v.clone() for k, v in sd.items()}, training=False, size=size)
    assert np.abs(ye.numpy() - g['y_eval']).max() <= 1e-4
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
               else v.clone()) for k, v in sd.items()}
    y = RT.discriminator_forward(x, sdr, training=True, size=size)
    (y * gy).sum().backward()
    assert np.abs(y.detach().numpy() - g['y_train']).max() <= 1e-4
    assert np.abs(sdr['classifier.0.bias'].grad.numpy() - g['g_classifier.0.bias']).max() <= 1e-4
    assert np.abs(sdr['features.0.weight'].grad.numpy() - g['g_features.0.weight']).max() <= 1e-3


def test_vgg(golden):
    g = golden('vgg')
    sd = synth.vgg19_state_dict(6, 34)
    x = synth.image_batch(6, 2, 3, 128, 128, name='vgg.x')
    gy = synth.normal_like(6, 'vgg.gy', (2, 512, 8, 8))
    xo = x.clone().requires_grad_(True)
    y = RT.vgg19_features_forward(xo, sd)
    (y * gy).sum().backward()
    assert np.abs(y.detach().numpy() - g['y']).max() <= 1e-4
    assert np.abs(xo.grad.numpy()[:, :, ::2, ::2] - g['gx_sub2']).max() <= 1e-3


def test_psnr(golden):
    g = golden('psnr')
    for i in range(3):
        a, b = torch.from_numpy(g['a%d' % i]), torch.from_numpy(g['b%d' % i])
        assert np.array_equal(RT.tensor2img(a), g['img_a%d' % i])
        assert abs(RT.psnr_sr(a, b, 4) - float(g['psnr%d' % i])) < 1e-9
