"""ORACLE support (test infrastructure) — generate ``tests/golden/*.npz`` by importing the REAL
reference (``/root/reference``) in the build container and running it on seeded synthetic
weights/inputs (``esrganplus_amd.synth``).  Also asserts that the functional restatement
``oracle/ref_torch.py`` reproduces the reference (this is what pins the oracle).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
The reference never travels; only the vectors written here are committed.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from esrganplus_amd import synth          # noqa: E402
from oracle import ref_import as RI       # noqa: E402
from oracle import ref_torch as RT        # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_grad_enabled(True)


def npy(t):
    return t.detach().cpu().numpy()


def checks(t):
    """(sum, abs-sum, L2) in float64 — compact whole-tensor checksums."""
    a = npy(t).astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])


def draw_z(seed, shapes, tag='z'):
    """Synthetic N(0,1) tensors standing in for GaussianNoise's normal_() draws (block.py:120).
    They are injected into the reference run with ``inject_z`` so the same numpy-generated z can
    be rebuilt on the GPU box (no dependence on torch's generator; nothing stored)."""
    return [synth.normal_like(seed, '%s.%d' % (tag, i), s) for i, s in enumerate(shapes)]


class inject_z:
    """Make the reference's ``self.noise.repeat(*x.size()).normal_()`` return our z tensors, in
    module execution order (one draw of the full activation shape per noise layer)."""

    def __init__(self, zs):
        self.zs = list(zs) if zs is not None else None

    def __enter__(self):
        self.orig = torch.Tensor.normal_
        if self.zs is None:
            return self
        it = iter(self.zs)

        def normal_(t, *a, **k):
            z = next(it)
            assert tuple(z.shape) == tuple(t.shape), (z.shape, t.shape)
            return t.copy_(z)
        torch.Tensor.normal_ = normal_
        self.it = it
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self.orig
        if self.zs is not None and exc[0] is None:
            assert next(self.it, None) is None, 'not all z consumed'


def assert_close(a, b, tol, what):
    d = (a - b).abs().max().item()
    print('  restatement vs reference  %-34s max|diff| = %.3e' % (what, d))
    assert d <= tol, (what, d)


# ----------------------------------------------------------------------------------------------
def gen_rdb():
    arch, blk = RI.codes_arch()
    sd = synth.rrdbnet_state_dict(nb=1, seed=11)
    p = 'model.1.sub.0.RDB1'
    with RI.cuda_to_cpu():
        m = blk.ResidualDenseBlock_5C(64)
    m.load_state_dict({k[len(p) + 1:]: v for k, v in sd.items() if k.startswith(p + '.')})
    x = synth.normal_like(11, 'rdb.x', (1, 64, 12, 12))
    gy = synth.normal_like(11, 'rdb.gy', (1, 64, 12, 12))
    res = {}
    for mode in ('eval', 'train'):
        m.train(mode == 'train')
        xr = x.clone().requires_grad_(True)
        m.zero_grad()
        z = draw_z(5, [x.shape], 'rdb.z')[0] if mode == 'train' else None
        with inject_z(None if z is None else [z]):
            y = m(xr)
        (y * gy).sum().backward()
        # restatement
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(p)}
        xo = x.clone().requires_grad_(True)
        yo = RT.rdb_forward(xo, sdr, p, z)
        (yo * gy).sum().backward()
        assert_close(y, yo, 1e-6, 'rdb %s fwd' % mode)
        assert_close(xr.grad, xo.grad, 1e-5, 'rdb %s grad_x' % mode)
        assert_close(m.conv5[0].weight.grad, sdr[p + '.conv5.0.weight'].grad, 1e-4,
                     'rdb %s grad conv5.w' % mode)
        res['y_' + mode] = npy(y)
        res['gx_' + mode] = npy(xr.grad)
        if mode == 'train':
            res['gw_conv1'] = npy(m.conv1[0].weight.grad)
            res['gw_conv3'] = npy(m.conv3[0].weight.grad)
            res['gw_conv5'] = npy(m.conv5[0].weight.grad)
            res['gb_conv4'] = npy(m.conv4[0].bias.grad)
            res['gw_conv1x1'] = npy(m.conv1x1.weight.grad)
    np.savez_compressed(os.path.join(OUT, 'rdb.npz'), **res)


def _run_net(net, sd_keys, x, gy, z):
    net.zero_grad()
    xr = x.clone().requires_grad_(True)
    with inject_z(z):
        y = net(xr)
    (y * gy).sum().backward()
    grads = {k: dict(net.named_parameters())[k].grad for k in sd_keys}
    return y, xr.grad, grads


def gen_rrdbnet_small():
    res = {}
    for tag, nb, shape, variant in (('a', 1, (1, 3, 16, 20), 'codes'),
                                    ('b', 2, (2, 3, 24, 24), 'codes'),
                                    ('c', 1, (1, 3, 13, 18), 'test_image')):
        sd = synth.rrdbnet_state_dict(nb=nb, seed=20 + nb)
        net = RI.build_rrdbnet(nb, variant)
        net.load_state_dict(sd, strict=True)
        x = synth.image_batch(3, *shape, name='small.x.' + tag)
        gy = synth.normal_like(3, 'small.gy.' + tag, (shape[0], 3, shape[2] * 4, shape[3] * 4))
        keys = list(sd.keys())
        for mode in ('eval', 'train'):
            net.train(mode == 'train')
            z = None
            if mode == 'train':
                z = draw_z(7, RT.noise_shapes(shape, nb, variant), 'small.z.' + tag)
            y, gx, grads = _run_net(net, keys, x, gy, z)
            sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            xo = x.clone().requires_grad_(True)
            yo = RT.rrdbnet_forward(xo, sdr, nb, z, variant)
            (yo * gy).sum().backward()
            assert_close(y, yo, 2e-6, 'rrdbnet[%s] %s fwd' % (tag, mode))
            assert_close(gx, xo.grad, 1e-4, 'rrdbnet[%s] %s grad_x' % (tag, mode))
            gmax = max((grads[k] - sdr[k].grad).abs().max().item() for k in keys)
            print('  restatement vs reference  rrdbnet[%s] %s param grads max|diff| = %.3e'
                  % (tag, mode, gmax))
            assert gmax < 2e-3
            res['%s_y_%s' % (tag, mode)] = npy(y)
            res['%s_gx_%s' % (tag, mode)] = npy(gx)
            res['%s_gchk_%s' % (tag, mode)] = np.stack([checks(grads[k]) for k in keys])
            if mode == 'train':
                for k in ('model.0.weight', 'model.1.sub.0.RDB2.conv2.0.weight',
                          'model.1.sub.0.RDB3.conv1x1.weight', 'model.1.sub.0.RDB1.conv5.0.bias',
                          'model.6.weight', 'model.10.weight', 'model.10.bias'):
                    res['%s_g_%s' % (tag, k)] = npy(grads[k])
    np.savez_compressed(os.path.join(OUT, 'rrdbnet_small.npz'), **res)


def gen_rrdbnet_full():
    from PIL import Image
    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    res = {}
    for variant in ('codes', 'test_image'):
        net = RI.build_rrdbnet(23, variant).eval()
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            x = synth.image_batch(0, 1, 3, 32, 32, name='full.x32')
            y = net(x)
            yo = RT.rrdbnet_forward(x, sd, 23, None, variant)
            assert_close(y, yo, 1e-5, 'rrdbnet nb=23 32x32 (%s)' % variant)
            if variant == 'codes':
                res['y32'] = npy(y)
                y32 = y
            else:
                assert_close(y, y32, 0.0, 'test_image copy == codes copy (eval)')
    # real image, reproducing test_image/test.py:26-40 with PIL instead of cv2
    img = np.array(Image.open(os.path.join(RI.REF, 'test_image', 'LR', 'baby.png')).convert('RGB'))
    x = torch.from_numpy(np.transpose(img.astype(np.float64) / 255, (2, 0, 1))).float()[None]
    with torch.no_grad():
        y = net(x)
        yo = RT.rrdbnet_forward(x, sd, 23, None)
    assert_close(y, yo, 1e-5, 'rrdbnet nb=23 baby.png 128x128')
    res['baby_lr_rgb'] = img
    res['baby_y_sub4'] = npy(y)[:, :, ::4, ::4]
    res['baby_y_chk'] = checks(y)
    out = y.squeeze().clamp(0, 1).numpy()
    res['baby_u8_sub4'] = (out * 255.0).round().astype(np.uint8)[:, ::4, ::4]
    print('  baby.png output range [%.3f, %.3f]' % (y.min().item(), y.max().item()))
    # odd-shaped natural image (57 wide x 86 tall) — the non-multiple-of-tile edge case
    img = np.array(Image.open(os.path.join(RI.REF, 'test_image', 'LR', 'woman.png')).convert('RGB'))
    print('  woman.png shape', img.shape)
    x = torch.from_numpy(np.transpose(img.astype(np.float64) / 255, (2, 0, 1))).float()[None]
    with torch.no_grad():
        y = net(x)
    res['woman_lr_rgb'] = img
    res['woman_y_sub4'] = npy(y)[:, :, ::4, ::4]
    res['woman_y_chk'] = checks(y)
    np.savez_compressed(os.path.join(OUT, 'rrdbnet_full.npz'), **res)


def grad_probe(key, g, nproj=3, nhead=32):
    """Compact view of one gradient tensor: (sum, abs-sum, L2) + `nproj` projections on fixed N(0,1) vectors
    (synth.normal_like(77, 'gproj.<key>.<i>')) + its first `nhead` entries — all float64."""
    a = npy(g).astype(np.float64).reshape(-1)
    pr = [float((a * synth.normal_like(77, 'gproj.%s.%d' % (key, i), a.shape).numpy().astype(np.float64)).sum())
          for i in range(nproj)]
    head = np.zeros(nhead)
    head[:min(nhead, a.size)] = a[:nhead]
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], pr, head])


def gen_rrdbnet_full_grad():
    """Full-depth backward (nb=23, the depth bench.py's fwd_bwd / gtrain time) of the imported reference on ONE
    128x128 LR tile, eval mode (GaussianNoise is the identity: block.py:117-123), loss = <y, gy>: every parameter
    gradient as a grad_probe row.  The GPU test runs the bench-shape batch (16 tiles) with this tile at two batch
    positions and zero upstream gradient elsewhere — parameter gradients are sums over the batch."""
    sd = synth.rrdbnet_state_dict(nb=23, seed=0, gain=0.5)
    net = RI.build_rrdbnet(23, 'codes').eval()
    net.load_state_dict(sd, strict=True)
    x = synth.image_batch(41, 1, 3, 128, 128, name='fullgrad.x')
    gy = synth.normal_like(41, 'fullgrad.gy', (1, 3, 512, 512)) / (3 * 512 * 512)
    keys = list(sd.keys())
    y, _, grads = _run_net(net, keys, x, gy, None)
    # the restatement agrees (forward + a sample of gradients) before anything is written
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = RT.rrdbnet_forward(x, sdr, 23, None, 'codes')
    (yo * gy).sum().backward()
    assert_close(y, yo, 1e-5, 'rrdbnet nb=23 128x128 fwd')
    rel = max(((grads[k] - sdr[k].grad).norm() / grads[k].norm().clamp_min(1e-30)).item() for k in keys)
    print('  restatement vs reference  rrdbnet nb=23 128x128 param grads max rel L2 diff = %.3e' % rel)
    assert rel < 1e-3
    res = {'y_chk': checks(y), 'y_head': npy(y)[0, :, :8, :8].astype(np.float64),
           'probe': np.stack([grad_probe(k, grads[k]) for k in keys]),
           'keys': np.array(keys)}
    for k in ('model.0.weight', 'model.0.bias', 'model.1.sub.0.RDB1.conv1.0.weight', 'model.1.sub.11.RDB2.conv1x1.weight',
              'model.1.sub.22.RDB3.conv5.0.bias', 'model.1.sub.22.RDB3.conv4.0.bias', 'model.1.sub.23.weight', 'model.10.weight'):
        res['g_' + k] = npy(grads[k])
    np.savez_compressed(os.path.join(OUT, 'rrdbnet_full_grad.npz'), **res)


class _Store16(torch.autograd.Function):
    """fp16 STORAGE of a tensor, emulated: the value is rounded to fp16 where the fp16 path writes it to memory, and so
    is the gradient that flows back through that point (times the loss scale, as the fp16 path keeps it)."""

    @staticmethod
    def forward(ctx, t, scale):
        ctx.scale = scale
        return t.half().float()

    @staticmethod
    def backward(ctx, g):
        s = ctx.scale
        return (g * s).half().float() / s, None


def rrdbnet_forward_fp16_storage(x, sd, nb, scale, z=None, variant='codes'):
    """oracle.ref_torch.rrdbnet_forward with every tensor the fp16 path keeps in memory rounded to fp16 — packed
    weights, block inputs / slices / outputs, head and tail activations — and fp32 accumulation inside each conv, which
    is what v_mfma_f32_32x32x16_f16 does.  NOT the GPU kernels' summation order: an independent statement of "fp16
    storage, fp32 accumulate" on the CPU.  z: the noise tensors of a training forward (None: eval)."""
    st = lambda t: _Store16.apply(t, scale)
    w = {k: (st(v) if k.endswith('.weight') else v) for k, v in sd.items()}      # biases stay fp32 (epilogue operands)
    conv, lrelu = RT._conv, RT._lrelu
    per = 3 if variant == 'codes' else 4
    fea = st(conv(st(x), w, 'model.0'))
    t = fea
    for i in range(nb):
        xr = t
        for j in (1, 2, 3):
            p = 'model.1.sub.%d.RDB%d' % (i, j)
            xb = t
            x1 = st(lrelu(conv(xb, w, p + '.conv1.0')))
            x2 = st(lrelu(conv(torch.cat((xb, x1), 1), w, p + '.conv2.0')) + F.conv2d(xb, w[p + '.conv1x1.weight']))
            x3 = st(lrelu(conv(torch.cat((xb, x1, x2), 1), w, p + '.conv3.0')))
            x4 = st(lrelu(conv(torch.cat((xb, x1, x2, x3), 1), w, p + '.conv4.0')) + x2)
            out = conv(torch.cat((xb, x1, x2, x3, x4), 1), w, p + '.conv5.0') * 0.2 + xb
            out = RT.gaussian_noise(out, None if z is None else z[per * i + j - 1])
            if j < 3:
                t = st(out)
            else:                                               # RDB3: the RRDB tail rides in the same epilogue
                t = st(RT.gaussian_noise(out * 0.2 + xr, None if (z is None or per == 3) else z[per * i + 3]))
    t = st(fea + conv(t, w, 'model.1.sub.%d' % nb))
    for key in ('model.3', 'model.6'):
        t = st(lrelu(conv(F.interpolate(t, scale_factor=2, mode='nearest'), w, key)))
    t = st(lrelu(conv(t, w, 'model.8')))
    return conv(t, w, 'model.10')


def gen_rrdbnet_full_grad_fp16emu():
    """How far does fp16 STORAGE alone move the nb = 23 parameter gradients from the fp32 reference's?  The restatement
    with fp16-rounded tensors (rrdbnet_forward_fp16_storage, loss scale 2^17 as the GPU test uses) against the
    imported reference's gradients in the GPU test's own metrics (relative L2, three random projections, the first 32
    entries; eight tensors in full).  The committed numbers are what tests/test_gpu_backward.py derives its fp16
    limits from (2 x these): the fp16 chains must sit where fp16 storage puts ANY implementation, not merely below a
    hand-picked bound."""
    import torch.nn.functional as F_  # noqa: F401
    sd = synth.rrdbnet_state_dict(nb=23, seed=0, gain=0.5)
    x = synth.image_batch(41, 1, 3, 128, 128, name='fullgrad.x')
    gy = synth.normal_like(41, 'fullgrad.gy', (1, 3, 512, 512)) / (3 * 512 * 512)
    keys = list(sd.keys())
    ref = dict(np.load(os.path.join(OUT, 'rrdbnet_full_grad.npz')))
    assert [str(k) for k in ref['keys']] == keys
    S = 2.0 ** 17
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = rrdbnet_forward_fp16_storage(x, sdr, 23, S)
    (y * gy).sum().backward()
    ey = float(np.abs(npy(y)[0, :, :8, :8] - ref['y_head']).max())
    errs = []
    for row, k in zip(ref['probe'], keys):
        got = sdr[k].grad.double().reshape(-1)
        l2 = float(row[2])
        e_l2 = abs(got.norm().item() - l2) / max(l2, 1e-30)
        e_pr = 0.0
        for i in range(3):
            r = synth.normal_like(77, 'gproj.%s.%d' % (k, i), tuple(got.shape)).double()
            e_pr = max(e_pr, abs((got * r).sum().item() - float(row[3 + i])) / max(l2, 1e-30))
        n = min(32, got.numel())
        head = row[6:6 + n]
        e_hd = np.abs(got[:n].numpy() - head).max() / max(np.abs(head).max(), l2 / np.sqrt(got.numel()))
        errs.append((e_l2, e_pr, e_hd))
    errs = np.array(errs)
    full_keys = [n[2:] for n in ref if n.startswith('g_')]
    full = np.array([np.abs(npy(sdr[k].grad) - ref['g_' + k]).max() / np.abs(ref['g_' + k]).max() for k in full_keys])
    print('  fp16-storage emulation vs fp32 reference: y_head err %.2e; worst (l2, proj, head) %s  mean %s'
          % (ey, errs.max(0), errs.mean(0)))
    for k, e in zip(full_keys, full):
        print('  %-40s max|diff| / max|ref| = %.2e' % (k, e))
    np.savez_compressed(os.path.join(OUT, 'rrdbnet_full_grad_fp16emu.npz'), worst=errs.max(0), mean=errs.mean(0),
                        worst_key=np.array([keys[i] for i in errs.argmax(0)]), y_head_err=np.array(ey),
                        full_keys=np.array(full_keys), full_err=full, loss_scale=np.array(S))


def gen_rrdbnet_small_fp16emu():
    """The fp16-storage emulation on the shape of tests/test_gpu_train_chain.py's oracle test (nb = 2, 2 x 3 x 24 x 40,
    GaussianNoise ON, both network copies): relative L2 distance of every parameter gradient from the fp32
    restatement's, same z.  Biases (sums with cancellation) sit at 4-5e-2 under fp16 storage whatever the
    implementation; the GPU test takes its limits from these numbers."""
    res = {}
    nb, shape = 2, (2, 3, 24, 40)
    sd = synth.rrdbnet_state_dict(nb=nb, seed=57)
    x = synth.image_batch(57, *shape, name='o16.x')
    gy = synth.normal_like(57, 'o16.gy', (shape[0], 3, 4 * shape[2], 4 * shape[3]))
    for variant in ('codes', 'test_image'):
        z = draw_z(58, RT.noise_shapes(shape, nb, variant), 'o16.z')
        a = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ya = RT.rrdbnet_forward(x, a, nb, z, variant)
        (ya * gy).sum().backward()
        b = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        yb = rrdbnet_forward_fp16_storage(x, b, nb, 1.0, z, variant)
        (yb * gy).sum().backward()
        errs = np.array([((b[k].grad - a[k].grad).norm() / a[k].grad.norm()).item() for k in sd])
        ey = ((ya - yb).abs().max() / (ya.max() - ya.min())).item()
        keys = list(sd.keys())
        print('  %s: output err / range %.2e, parameter-gradient rel L2: worst %.3e (%s), mean %.3e'
              % (variant, ey, errs.max(), keys[int(errs.argmax())], errs.mean()))
        res[variant + '_y'] = np.array(ey)
        res[variant + '_worst'] = np.array(errs.max())
        res[variant + '_mean'] = np.array(errs.mean())
    np.savez_compressed(os.path.join(OUT, 'rrdbnet_small_fp16emu.npz'), **res)


def gen_disc():
    sd = synth.discriminator_state_dict(seed=4)
    net = RI.build_discriminator()
    net.load_state_dict(sd, strict=True)
    x = synth.image_batch(4, 4, 3, 128, 128, name='disc.x')
    gy = synth.normal_like(4, 'disc.gy', (4, 1))
    res = {}
    # eval forward
    net.eval()
    with torch.no_grad():
        ye = net(x)
        yo = RT.discriminator_forward(x, {k: v.clone() for k, v in sd.items()}, training=False)
    assert_close(ye, yo, 1e-4, 'D eval fwd')
    res['y_eval'] = npy(ye)
    # train forward + backward, then 3 more forwards (running stats after 4 calls)
    net.train()
    xr = x.clone().requires_grad_(True)
    y = net(xr)
    (y * gy).sum().backward()
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
               else v.clone()) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    yo = RT.discriminator_forward(xo, sdr, training=True)
    (yo * gy).sum().backward()
    assert_close(y, yo, 1e-4, 'D train fwd')
    assert_close(xr.grad, xo.grad, 1e-4, 'D train grad_x')
    params = dict(net.named_parameters())
    gmax = max((params[k].grad - sdr[k].grad).abs().max().item() for k in params)
    print('  restatement vs reference  D param grads max|diff| = %.3e' % gmax)
    res['y_train'] = npy(y)
    res['gx_chk'] = checks(xr.grad)
    res['gx_sub8'] = npy(xr.grad)[:, :, ::8, ::8]
    keys = list(params.keys())
    res['gchk'] = np.stack([checks(params[k].grad) for k in keys])
    for k in ('features.0.weight', 'features.3.weight', 'features.3.bias', 'features.26.bias',
              'features.27.weight', 'classifier.2.weight', 'classifier.0.bias'):
        res['g_' + k] = npy(params[k].grad)
    res['g_features.2.weight_sub'] = npy(params['features.2.weight'].grad)[::4, ::4]
    with torch.no_grad():
        for i in range(3):
            net(x * (0.5 + 0.25 * i))
            RT.discriminator_forward(x * (0.5 + 0.25 * i), sdr, training=True)
    bufs = dict(net.named_buffers())
    for k in ('features.3', 'features.15', 'features.27'):
        assert_close(bufs[k + '.running_var'], sdr[k + '.running_var'], 1e-5, 'D ' + k + ' rv')
        res['rm_' + k] = npy(bufs[k + '.running_mean'])
        res['rv_' + k] = npy(bufs[k + '.running_var'])
    res['nbt'] = npy(bufs['features.3.num_batches_tracked'])
    np.savez_compressed(os.path.join(OUT, 'disc.npz'), **res)


def gen_disc_sn():
    """Discriminator_VGG_128_SN (architecture.py:131-175, spectral_norm.py): two training forwards (each runs one
    power iteration), the first one's backward, then an eval forward on the buffers they left."""
    sd = synth.discriminator_sn_state_dict(seed=6)
    net = RI.build_discriminator_sn()
    net.load_state_dict(sd, strict=True)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    x = synth.image_batch(6, 3, 3, 128, 128, name='dsn.x')
    gy = synth.normal_like(6, 'dsn.gy', (3, 1))
    net.train()
    xr = x.clone().requires_grad_(True)
    y1 = net(xr)
    (y1 * gy).sum().backward()
    params = dict(net.named_parameters())
    res = {'y1': npy(y1), 'gx_chk': checks(xr.grad), 'gx_sub8': npy(xr.grad)[:, :, ::8, ::8],
           'keys': np.array(sorted(params.keys())),
           'gchk': np.stack([checks(params[k].grad) for k in sorted(params.keys())])}
    for k in ('conv0.weight_orig', 'conv0.bias', 'conv3.bias', 'linear1.weight_orig', 'linear0.bias'):
        res['g_' + k] = npy(params[k].grad)
    res['g_conv5.weight_orig_sub'] = npy(params['conv5.weight_orig'].grad)[::8, ::8]
    with torch.no_grad():
        y2 = net(x * 0.75)
    res['y2'] = npy(y2)
    bufs = dict(net.named_buffers())
    for k in ('conv0', 'conv4', 'conv9', 'linear0', 'linear1'):
        res['u_' + k] = npy(bufs[k + '.weight_u'])
        res['wchk_' + k] = checks(bufs[k + '.weight'])
    net.eval()
    with torch.no_grad():
        res['y_eval'] = npy(net(x))
    np.savez_compressed(os.path.join(OUT, 'disc_sn.npz'), **res)


def gen_disc_sn_two():
    """Discriminator_VGG_128_SN in the train step's call order (SRRaGAN_model.py:150-167): TWO training forwards — each
    with its own power iteration, so each with its own normalised weights — and only then the backward of both.  Pins
    that a call's backward uses the weights of ITS forward."""
    sd = synth.discriminator_sn_state_dict(seed=8)
    net = RI.build_discriminator_sn()
    net.load_state_dict(sd, strict=True)
    xa = synth.image_batch(8, 2, 3, 128, 128, name='dsn2.xa')
    xb = synth.image_batch(9, 2, 3, 128, 128, name='dsn2.xb')
    ga = synth.normal_like(8, 'dsn2.ga', (2, 1))
    gb = synth.normal_like(9, 'dsn2.gb', (2, 1))
    net.train()
    xar = xa.clone().requires_grad_(True)
    ya = net(xar)
    yb = net(xb)
    ((ya * ga).sum() + (yb * gb).sum()).backward()
    params = dict(net.named_parameters())
    res = {'ya': npy(ya), 'yb': npy(yb), 'gx_chk': checks(xar.grad), 'gx_sub8': npy(xar.grad)[:, :, ::8, ::8],
           'keys': np.array(sorted(params.keys())),
           'gchk': np.stack([checks(params[k].grad) for k in sorted(params.keys())])}
    for k in ('conv0.weight_orig', 'conv9.bias', 'linear1.weight_orig', 'linear0.bias'):
        res['g_' + k] = npy(params[k].grad)
    np.savez_compressed(os.path.join(OUT, 'disc_sn_two.npz'), **res)


def gen_disc_variants():
    """Discriminator_VGG_96 / _192 (architecture.py:178-270): eval + train forward, gradient checksums."""
    for size, batch in ((96, 3), (192, 2)):
        sd = synth.discriminator_state_dict(seed=40 + size, size=size)
        net = RI.build_discriminator(size)
        net.load_state_dict(sd, strict=True)
        x = synth.image_batch(size, batch, 3, size, size, name='disc%d.x' % size)
        gy = synth.normal_like(size, 'disc%d.gy' % size, (batch, 1))
        res = {}
        net.eval()
        with torch.no_grad():
            ye = net(x)
            yo = RT.discriminator_forward(x, {k: v.clone() for k, v in sd.items()}, training=False, size=size)
        assert_close(ye, yo, 1e-4, 'D%d eval fwd' % size)
        res['y_eval'] = npy(ye)
        net.train()
        xr = x.clone().requires_grad_(True)
        y = net(xr)
        (y * gy).sum().backward()
        sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
                   else v.clone()) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        yo = RT.discriminator_forward(xo, sdr, training=True, size=size)
        (yo * gy).sum().backward()
        assert_close(y, yo, 1e-4, 'D%d train fwd' % size)
        assert_close(xr.grad, xo.grad, 1e-4, 'D%d train grad_x' % size)
        params = dict(net.named_parameters())
        res['y_train'] = npy(y)
        res['gx_chk'] = checks(xr.grad)
        res['gchk'] = np.stack([checks(params[k].grad) for k in params])
        last = [k for k in params if k.startswith('features.') and k.endswith('.weight')][-1]
        res['g_last_bn_weight'] = npy(params[last].grad)
        res['g_classifier.0.bias'] = npy(params['classifier.0.bias'].grad)
        res['g_features.0.weight'] = npy(params['features.0.weight'].grad)
        np.savez_compressed(os.path.join(OUT, 'disc%d.npz' % size), **res)


def gen_srresnet():
    """SRResNet x4 (architecture.py:13-44; nb=3 here) with both upsamplers: forward, input gradient, every
    parameter gradient's checksum and three full ones — straight from the imported reference."""
    for mode in ('pixelshuffle', 'upconv'):
        nb = 3
        sd = synth.srresnet_state_dict(nb=nb, seed=77, upsample_mode=mode)
        net = RI.build_srresnet(nb, mode)
        net.load_state_dict(sd, strict=True)
        x = synth.image_batch(77, 2, 3, 20, 28, name='srresnet.x')
        gy = synth.normal_like(77, 'srresnet.gy', (2, 3, 80, 112))
        xr = x.clone().requires_grad_(True)
        y = net(xr)
        (y * gy).sum().backward()
        params = dict(net.named_parameters())
        res = dict(y=npy(y), gx=npy(xr.grad), gchk=np.stack([checks(params[k].grad) for k in params]))
        for key in ('model.0.weight', 'model.1.sub.1.res.2.bias', 'model.10.weight'):
            res['g_' + key] = npy(params[key].grad)
        up = 'model.2.weight' if mode == 'pixelshuffle' else 'model.3.weight'
        res['g_up'] = npy(params[up].grad)
        np.savez_compressed(os.path.join(OUT, 'srresnet_%s.npz' % mode), **res)


def _vgg19_features():
    layers, cin = [], 3
    for v in synth.VGG19_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def install_vgg_stub(seed):
    """torchvision is absent; give the stub a ``vgg19`` whose ``features`` is cfg 'E' (the
    published torchvision layout) loaded with synthetic He weights.  VGG19 parity is therefore
    NOT pinned by the reference (no weights, no module) — see DESIGN.md."""
    RI._stub_torchvision()

    def vgg19(pretrained=False):
        m = types.SimpleNamespace()
        f = _vgg19_features()
        f.load_state_dict({k[len('features.'):]: v
                           for k, v in synth.vgg19_state_dict(seed, 36).items()}, strict=True)
        m.features = f
        return m
    sys.modules['torchvision'].models.vgg19 = vgg19
    sys.modules['torchvision.models'].vgg19 = vgg19


def gen_vgg():
    install_vgg_stub(6)
    arch, _ = RI.codes_arch()
    netF = arch.VGGFeatureExtractor(feature_layer=34, use_bn=False, use_input_norm=True).eval()
    sd = synth.vgg19_state_dict(6, 34)
    x = synth.image_batch(6, 2, 3, 128, 128, name='vgg.x')
    gy = synth.normal_like(6, 'vgg.gy', (2, 512, 8, 8))
    xr = x.clone().requires_grad_(True)
    y = netF(xr)
    (y * gy).sum().backward()
    xo = x.clone().requires_grad_(True)
    yo = RT.vgg19_features_forward(xo, sd)
    (yo * gy).sum().backward()
    assert_close(y, yo, 1e-4, 'VGG19[:35] fwd')
    assert_close(xr.grad, xo.grad, 1e-3, 'VGG19[:35] grad_x')
    print('  VGG feature magnitude: mean|y| = %.3f' % y.abs().mean().item())
    np.savez_compressed(os.path.join(OUT, 'vgg.npz'), y=npy(y),
                        gx_sub2=npy(xr.grad)[:, :, ::2, ::2], gx_chk=checks(xr.grad))


def gen_train_step():
    """One real SRRaGANModel.optimize_parameters step (SRRaGAN_model.py:113-186), nb=2, batch 4."""
    install_vgg_stub(6)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    RI.codes_arch()
    from models import create_model
    opt = {'model': 'srragan', 'scale': 4, 'gpu_ids': None, 'is_train': True,
           'path': {'pretrain_model_G': None, 'pretrain_model_D': None},
           'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': 64,
                         'nb': 2, 'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4},
           'network_D': {'which_model_D': 'discriminator_vgg_128', 'norm_type': 'batch',
                         'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64, 'in_nc': 3},
           'train': {'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4,
                     'weight_decay_D': 0, 'beta1_D': 0.9, 'lr_scheme': 'MultiStepLR',
                     'lr_steps': [50000, 100000, 200000, 300000], 'lr_gamma': 0.5,
                     'pixel_criterion': 'l1', 'pixel_weight': 0.01, 'feature_criterion': 'l1',
                     'feature_weight': 1, 'gan_type': 'vanilla', 'gan_weight': 0.005,
                     'D_update_ratio': None, 'D_init_iters': None}}
    with RI.cuda_to_cpu():
        model = create_model(opt)
    sdG = synth.rrdbnet_state_dict(nb=2, seed=30)
    sdD = synth.discriminator_state_dict(seed=31)
    model.netG.load_state_dict(sdG, strict=True)
    model.netD.load_state_dict(sdD, strict=True)
    lr = synth.image_batch(30, 4, 3, 32, 32, name='step.lr')
    hr = synth.image_batch(30, 4, 3, 128, 128, name='step.hr')
    model.feed_data({'LR': lr, 'HR': hr})
    z = draw_z(9, RT.noise_shapes(lr.shape, 2, 'codes'), 'step.z')
    with inject_z(z):
        model.optimize_parameters(1)
    log = model.get_current_log()
    res = {}
    for k, v in log.items():
        res['log_' + k] = np.array(float(v))
        print('  %-10s %.6e' % (k, float(v)))
    res['fake_H_chk'] = checks(model.fake_H)
    res['fake_H_sub4'] = npy(model.fake_H)[:, :, ::4, ::4]
    g = dict(model.netG.named_parameters())
    d = dict(model.netD.named_parameters())
    res['G_new_chk'] = np.stack([checks(g[k]) for k in sdG.keys()])
    res['D_new_chk'] = np.stack([checks(d[k]) for k in d.keys()])
    res['G_delta_model.0.weight'] = npy(g['model.0.weight'] - sdG['model.0.weight'])
    res['D_delta_classifier.2.weight'] = npy(d['classifier.2.weight'] - sdD['classifier.2.weight'])
    np.savez_compressed(os.path.join(OUT, 'train_step.npz'), **res)


def _srragan_opt(nb):
    return {'model': 'srragan', 'scale': 4, 'gpu_ids': None, 'is_train': True,
            'path': {'pretrain_model_G': None, 'pretrain_model_D': None},
            'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': 64,
                          'nb': nb, 'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4},
            'network_D': {'which_model_D': 'discriminator_vgg_128', 'norm_type': 'batch',
                          'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64, 'in_nc': 3},
            'train': {'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4,
                      'weight_decay_D': 0, 'beta1_D': 0.9, 'lr_scheme': 'MultiStepLR',
                      'lr_steps': [50000, 100000, 200000, 300000], 'lr_gamma': 0.5,
                      'pixel_criterion': 'l1', 'pixel_weight': 0.01, 'feature_criterion': 'l1',
                      'feature_weight': 1, 'gan_type': 'vanilla', 'gan_weight': 0.005,
                      'D_update_ratio': None, 'D_init_iters': None}}


class _GEmu16(nn.Module):
    """The reference's netG slot filled with the fp16-storage restatement of its forward (rrdbnet_forward_fp16_storage)
    over the SAME nn.Parameters (the model's Adam keeps updating them)."""

    def __init__(self, inner, nb, scale, zs):
        super().__init__()
        self.inner, self.nb, self.scale, self.zs = inner, nb, scale, zs

    def forward(self, x):
        sd = dict(self.inner.named_parameters())
        return rrdbnet_forward_fp16_storage(x, sd, self.nb, self.scale, self.zs if self.training else None)


class _SeqEmu16(nn.Module):
    """netD / netF under fp16 storage: conv / linear weights rounded to fp16 (the packed copies), every tensor a layer
    writes to memory — conv, BatchNorm, activation, pool, linear outputs — rounded to fp16, gradients through those
    points rounded at the loss scale; arithmetic inside a layer in fp32 (what an fp32-accumulating kernel does)."""

    def __init__(self, inner, scale):
        super().__init__()
        self.inner, self.scale = inner, scale
        for m in inner.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.Linear, nn.LeakyReLU, nn.ReLU, nn.MaxPool2d)):
                m.register_forward_hook(lambda mod, inp, out: _Store16.apply(out, scale))

    def forward(self, x):
        from torch.func import functional_call
        st = {k: (_Store16.apply(v, self.scale) if (k.endswith('.weight') and v.dim() >= 2) else v)
              for k, v in self.inner.named_parameters()}
        st.update(dict(self.inner.named_buffers()))
        return functional_call(self.inner, st, (_Store16.apply(x, self.scale),))


def gen_train_step_full():
    """ONE real SRRaGANModel.optimize_parameters step (SRRaGAN_model.py:113-186) at the benchmarked DEPTH — nb = 23,
    batch 4 of 32x32 LR crops — so that the composition (loss plumbing, accumulation into dL/d fake_H, stream order,
    loss scaling) is pinned at depth and not only at nb = 2 (VERDICT r04 missing #4).  Also the same step under an
    fp16-STORAGE emulation of all three networks (loss scale 1024): how far fp16 storage alone moves the seven logged
    losses and the first Adam update — the GPU test's fp16 limits are 2 x these distances."""
    install_vgg_stub(6)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    RI.codes_arch()
    from models import create_model
    nb = 23
    sdG = synth.rrdbnet_state_dict(nb=nb, seed=60, gain=0.5)
    sdD = synth.discriminator_state_dict(seed=61)
    lr = synth.image_batch(62, 4, 3, 32, 32, name='stepfull.lr')
    hr = synth.image_batch(62, 4, 3, 128, 128, name='stepfull.hr')
    z = draw_z(63, RT.noise_shapes(lr.shape, nb, 'codes'), 'stepfull.z')
    keys = ('l_g_pix', 'l_g_fea', 'l_g_gan', 'l_d_real', 'l_d_fake', 'D_real', 'D_fake')
    probes = ('model.0.weight', 'model.1.sub.0.RDB1.conv1.0.weight', 'model.1.sub.11.RDB2.conv5.0.weight',
              'model.1.sub.22.RDB3.conv1x1.weight', 'model.1.sub.23.weight', 'model.10.weight')

    def run(emu):
        with RI.cuda_to_cpu():
            model = create_model(_srragan_opt(nb))
        model.netG.load_state_dict(sdG, strict=True)
        model.netD.load_state_dict(sdD, strict=True)
        if emu:
            S = 1024.0
            model.netG = _GEmu16(model.netG, nb, S, z).train()
            model.netD = _SeqEmu16(model.netD, S).train()
            model.netF = _SeqEmu16(model.netF, S).eval()
        model.feed_data({'LR': lr, 'HR': hr})
        if emu:
            model.optimize_parameters(1)
        else:
            with inject_z(z):
                model.optimize_parameters(1)
        log = model.get_current_log()
        g = dict((model.netG.inner if emu else model.netG).named_parameters())
        d = dict((model.netD.inner if emu else model.netD).named_parameters())
        return model, np.array([float(log[k]) for k in keys]), g, d

    model, logv, g, d = run(False)
    res = {'log': logv, 'keys': np.array(keys)}
    for k, v in zip(keys, logv):
        print('  %-10s %.6e' % (k, v))
    res['fake_H_chk'] = checks(model.fake_H)
    res['fake_H_sub4'] = npy(model.fake_H)[:, :, ::4, ::4]
    res['G_new_chk'] = np.stack([checks(g[k]) for k in sdG.keys()])
    res['D_new_chk'] = np.stack([checks(d[k]) for k in d.keys()])
    for k in probes:
        res['G_delta_' + k] = npy(g[k] - sdG[k])
    for k in ('classifier.2.weight', 'features.0.weight', 'features.26.weight'):
        res['D_delta_' + k] = npy(d[k] - sdD[k])[:8]          # (the first 8 output channels: features.26 is 4 M weights)
    model16, log16, g16, d16 = run(True)
    res['log_fp16emu'] = log16
    res['log_fp16emu_err'] = np.abs(log16 - logv)
    print('  fp16-storage emulation of the step: |loss - fp32 loss| =', res['log_fp16emu_err'])
    print('                                       fp32 losses       =', logv)
    # the first Adam update is ~lr * sign(gradient): fraction of entries whose update has the reference's sign
    res['G_sign_agree_fp16emu'] = np.array([np.mean(np.sign(npy(g16[k] - sdG[k])) == np.sign(res['G_delta_' + k])) for k in probes])
    res['D_sign_agree_fp16emu'] = np.array([np.mean(np.sign(npy(d16[k] - sdD[k])[:8]) == np.sign(res['D_delta_' + k]))
                                            for k in ('classifier.2.weight', 'features.0.weight', 'features.26.weight')])
    res['probes'] = np.array(probes)
    fh = npy(model.fake_H)
    res['fake_H_fp16emu_err'] = np.array(np.abs(npy(model16.fake_H) - fh).max() / (fh.max() - fh.min()))
    print('  sign agreement of the first Adam update under fp16 storage: G %s  D %s; fake_H err / range %.2e'
          % (res['G_sign_agree_fp16emu'], res['D_sign_agree_fp16emu'], float(res['fake_H_fp16emu_err'])))
    np.savez_compressed(os.path.join(OUT, 'train_step_full.npz'), **res)


def gen_sr_infer():
    """test_image/test.py:26-40 end to end on ALL five bundled LR images (128x128, 72x72, 64x64, 70x70, 57x86) with the
    imported `RRDB_Net` and the synthetic nb = 23 weights tools/sr_infer.py loads for `synthetic`: the uint8 images the
    script would write (RGB order; cv2 is absent, PIL reads the same pixels).  The LR pixels travel in the fixture —
    they are the reference's test data — so the GPU box can re-create the input directory."""
    from PIL import Image
    import glob
    arch, _ = RI.test_image_arch()
    with RI.cuda_to_cpu():
        model = arch.RRDB_Net(3, 3, 64, 23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu',
                              mode='CNA', res_scale=1, upsample_mode='upconv')
    model.load_state_dict(synth.rrdbnet_state_dict(23, 0), strict=False)
    model.eval()
    res = {}
    names = []
    for path in sorted(glob.glob(os.path.join(RI.REF, 'test_image', 'LR', '*'))):
        base = os.path.splitext(os.path.basename(path))[0]
        rgb = np.array(Image.open(path).convert('RGB'))
        img = torch.from_numpy(np.transpose(rgb * 1.0 / 255, (2, 0, 1))).float().unsqueeze(0)
        with torch.no_grad():
            out = model(img).data.squeeze().float().cpu().clamp_(0, 1).numpy()
        out = (np.transpose(out, (1, 2, 0)) * 255.0).round().astype(np.uint8)
        # distance of every value from its rounding boundary: how many pixels CAN flip by 1 LSB under 1e-4 of error
        names.append(base)
        res['lr_' + base], res['sr_' + base] = rgb, out
        print('  %-10s %s -> %s  mean %.1f  saturated %.1f %%' % (base, rgb.shape[:2], out.shape[:2], out.mean(),
                                                                 100 * np.mean((out == 0) | (out == 255))))
    res['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'sr_infer.npz'), **res)


def gen_train_steps3():
    """THREE iterations of the reference's training loop body (codes/train.py:97-106: ``update_learning_rate()`` —
    the schedulers are stepped BEFORE the optimizers — then ``feed_data`` + ``optimize_parameters``) on the real
    SRRaGANModel, nb=2, batch 4, MultiStepLR([1, 2], gamma 0.5), fresh data and noise per step: pins Adam's moments /
    bias correction over several steps, the scheduler order, and the BatchNorm running statistics after 12 netD
    forwards (4 per step, SRRaGAN_model.py:133-134,150-151)."""
    install_vgg_stub(6)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    RI.codes_arch()
    from models import create_model
    opt = {'model': 'srragan', 'scale': 4, 'gpu_ids': None, 'is_train': True,
           'path': {'pretrain_model_G': None, 'pretrain_model_D': None},
           'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': 64,
                         'nb': 2, 'in_nc': 3, 'out_nc': 3, 'gc': 32, 'scale': 4},
           'network_D': {'which_model_D': 'discriminator_vgg_128', 'norm_type': 'batch',
                         'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64, 'in_nc': 3},
           'train': {'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4,
                     'weight_decay_D': 0, 'beta1_D': 0.9, 'lr_scheme': 'MultiStepLR',
                     'lr_steps': [1, 2], 'lr_gamma': 0.5,
                     'pixel_criterion': 'l1', 'pixel_weight': 0.01, 'feature_criterion': 'l1',
                     'feature_weight': 1, 'gan_type': 'vanilla', 'gan_weight': 0.005,
                     'D_update_ratio': None, 'D_init_iters': None}}
    with RI.cuda_to_cpu():
        model = create_model(opt)
    sdG = synth.rrdbnet_state_dict(nb=2, seed=32)
    sdD = synth.discriminator_state_dict(seed=33)
    model.netG.load_state_dict(sdG, strict=True)
    model.netD.load_state_dict(sdD, strict=True)
    res = {}
    import warnings
    for it in range(1, 4):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')                  # "lr_scheduler.step() before optimizer.step()": the reference's order
            model.update_learning_rate()
        lr = synth.image_batch(70 + it, 4, 3, 32, 32, name='steps3.lr')
        hr = synth.image_batch(80 + it, 4, 3, 128, 128, name='steps3.hr')
        model.feed_data({'LR': lr, 'HR': hr})
        z = draw_z(90 + it, RT.noise_shapes(lr.shape, 2, 'codes'), 'steps3.z')
        with inject_z(z):
            model.optimize_parameters(it)
        log = model.get_current_log()
        res['lr_%d' % it] = np.array([model.optimizer_G.param_groups[0]['lr'], model.optimizer_D.param_groups[0]['lr']])
        res['log_%d' % it] = np.array([float(log[k]) for k in ('l_g_pix', 'l_g_fea', 'l_g_gan', 'l_d_real', 'l_d_fake',
                                                              'D_real', 'D_fake')])
        print('  step %d lr %s log %s' % (it, res['lr_%d' % it], res['log_%d' % it]))
        res['fake_H_chk_%d' % it] = checks(model.fake_H)
    g = dict(model.netG.named_parameters())
    d = dict(model.netD.named_parameters())
    res['G_chk'] = np.stack([checks(g[k]) for k in sdG.keys()])
    res['D_chk'] = np.stack([checks(d[k]) for k in d.keys()])
    res['G_delta_model.0.weight'] = npy(g['model.0.weight'] - sdG['model.0.weight'])
    res['G_delta_model.1.sub.1.RDB2.conv3.0.bias'] = npy(g['model.1.sub.1.RDB2.conv3.0.bias'] - sdG['model.1.sub.1.RDB2.conv3.0.bias'])
    res['D_delta_classifier.2.weight'] = npy(d['classifier.2.weight'] - sdD['classifier.2.weight'])
    res['D_delta_features.3.weight'] = npy(d['features.3.weight'] - sdD['features.3.weight'])
    bufs = dict(model.netD.named_buffers())
    for k in ('features.3', 'features.15', 'features.27'):
        res['rm_' + k] = npy(bufs[k + '.running_mean'])
        res['rv_' + k] = npy(bufs[k + '.running_var'])
    res['nbt'] = np.array([int(bufs[k]) for k in bufs if k.endswith('num_batches_tracked')])
    print('  num_batches_tracked', res['nbt'])
    np.savez_compressed(os.path.join(OUT, 'train_steps3.npz'), **res)


def gen_psnr():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    RI._stub_torchvision()
    p = os.path.join(RI.REF, 'codes')
    if p not in sys.path:
        sys.path.insert(0, p)
    import importlib
    util = importlib.import_module('utils.util')
    res = {}
    for i, (h, w) in enumerate(((32, 40), (64, 64), (17, 23))):
        a = synth.image_batch(40 + i, 1, 3, h, w, name='psnr.a')[0] * 1.2 - 0.1
        b = (a + 0.03 * synth.normal_like(40 + i, 'psnr.n', (3, h, w)))
        ia, ib = util.tensor2img(a.clone()), util.tensor2img(b.clone())
        assert (ia == RT.tensor2img(a)).all()
        ca = (ia / 255.)[4:-4, 4:-4, :] * 255
        cb = (ib / 255.)[4:-4, 4:-4, :] * 255
        ps = util.calculate_psnr(ca, cb)
        assert abs(ps - RT.psnr_sr(a, b, 4)) < 1e-12
        res['a%d' % i], res['b%d' % i] = npy(a), npy(b)
        res['img_a%d' % i] = ia
        res['psnr%d' % i] = np.array(ps)
        print('  psnr case %d: %.4f dB' % (i, ps))
    np.savez_compressed(os.path.join(OUT, 'psnr.npz'), **res)


def gen_metrics():
    """Y-channel conversion (codes/data/util.py:123-168) from the imported reference.  (SSIM,
    codes/utils/util.py:117-158: gen_ssim below, the reference's own lines over a two-function cv2 shim.)"""
    U = RI.data_util()
    rng = np.random.RandomState(7)
    u8 = rng.randint(0, 256, (12, 10, 3)).astype(np.uint8)
    fl = rng.rand(12, 10, 3).astype(np.float32)
    res = {'u8': u8, 'fl': fl}
    for name, fn in (('bgr', U.bgr2ycbcr), ('rgb', U.rgb2ycbcr)):
        for only_y in (True, False):
            res['%s_u8_%d' % (name, only_y)] = fn(u8.copy(), only_y)
            res['%s_fl_%d' % (name, only_y)] = fn(fl.copy(), only_y)
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **res)


def gen_metrics_y():
    """PSNR on the Y channel exactly as the reference's test script forms it (codes/test.py:69-90): tensor2img both
    images, /255 (float64), bgr2ycbcr(only_y) on the FLOAT images (the unrounded branch of data/util.py:150-168),
    crop, x255, calculate_psnr.  (SSIM / SSIM_Y of the same flow: gen_ssim.)"""
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    RI._stub_torchvision()
    p = os.path.join(RI.REF, 'codes')
    if p not in sys.path:
        sys.path.insert(0, p)
    import importlib
    util = importlib.import_module('utils.util')
    U = RI.data_util()
    res = {}
    for i, (h, w, crop) in enumerate(((40, 52, 4), (33, 47, 2), (128, 96, 4))):
        hr = synth.image_batch(80 + i, 1, 3, h, w, name='mety.hr')[0]
        sr = hr + 0.06 * synth.normal_like(80 + i, 'mety.n', (3, h, w))
        sr_img = util.tensor2img(sr.clone()) / 255.
        gt_img = util.tensor2img(hr.clone()) / 255.
        c_sr, c_gt = sr_img[crop:-crop, crop:-crop, :], gt_img[crop:-crop, crop:-crop, :]
        psnr = util.calculate_psnr(c_sr * 255, c_gt * 255)
        sr_y, gt_y = U.bgr2ycbcr(sr_img, only_y=True), U.bgr2ycbcr(gt_img, only_y=True)     # (scales its input in place)
        psnr_y = util.calculate_psnr(sr_y[crop:-crop, crop:-crop] * 255, gt_y[crop:-crop, crop:-crop] * 255)
        # inputs are seeded (synth): the tests regenerate them; kept: sizes, results, one Y plane
        res['shape%d' % i], res['crop%d' % i] = np.array([h, w]), np.array(crop)
        res['psnr%d' % i], res['psnr_y%d' % i] = np.array(psnr), np.array(psnr_y)
        if i == 0:
            res['y_sr0'] = (sr_y * 255).astype(np.float64)
        print('  metrics_y case %d: PSNR %.6f dB, PSNR_Y %.6f dB' % (i, psnr, psnr_y))
    np.savez_compressed(os.path.join(OUT, 'metrics_y.npz'), **res)


def gen_ssim():
    """SSIM from the reference's OWN lines (codes/utils/util.py:117-158: ``ssim``, ``calculate_ssim`` incl. the
    three-fold loop over the full (H, W, 3) arrays at 152-153), run over ``ref_import.cv2_shim`` (cv2 is absent: the
    shim supplies OpenCV's getGaussianKernel / filter2D semantics, nothing of the reference's).  Cases: the
    reference's validation flow (codes/test.py:69-90: tensor2img, /255, crop, x255 -> SSIM on 3 channels and on the
    unrounded Y plane), a 2-D plane, an (H, W, 1) image, and raw float planes.  Inputs are seeded (synth): the
    tests regenerate them; the fixture keeps sizes, results and one SSIM input pair in full."""
    util = RI.utils_util()
    U = RI.data_util()
    res = {}
    for i, (h, w, crop) in enumerate(((40, 52, 4), (33, 47, 2), (128, 96, 4), (30, 27, 0))):
        hr = synth.image_batch(90 + i, 1, 3, h, w, name='ssim.hr')[0]
        sr = hr + 0.06 * synth.normal_like(90 + i, 'ssim.n', (3, h, w))
        sr_img = util.tensor2img(sr.clone()) / 255.
        gt_img = util.tensor2img(hr.clone()) / 255.
        sl = slice(crop, -crop) if crop else slice(None)
        c_sr, c_gt = sr_img[sl, sl, :], gt_img[sl, sl, :]
        ssim = util.calculate_ssim(c_sr * 255, c_gt * 255)
        sr_y, gt_y = U.bgr2ycbcr(sr_img, only_y=True), U.bgr2ycbcr(gt_img, only_y=True)    # scales its input in place
        ssim_y = util.calculate_ssim(sr_y[sl, sl] * 255, gt_y[sl, sl] * 255)
        res['shape%d' % i], res['crop%d' % i] = np.array([h, w]), np.array(crop)
        res['ssim%d' % i], res['ssim_y%d' % i] = np.array(ssim), np.array(ssim_y)
        print('  ssim case %d: SSIM %.12f  SSIM_Y %.12f' % (i, ssim, ssim_y))
    # grey: a 1-channel tensor -> tensor2img gives (H, W); and the (H, W, 1) branch of calculate_ssim
    g_hr = synth.image_batch(95, 1, 1, 24, 30, name='ssim.g')[0]
    g_sr = g_hr + 0.05 * synth.normal_like(95, 'ssim.gn', (1, 24, 30))
    ia, ib = util.tensor2img(g_sr.clone()), util.tensor2img(g_hr.clone())
    res['grey'] = np.array(util.calculate_ssim(ia[2:-2, 2:-2].astype(np.float64), ib[2:-2, 2:-2].astype(np.float64)))
    res['grey_hw1'] = np.array(util.calculate_ssim(ia[..., None], ib[..., None]))
    # raw float planes / 3-channel arrays in full (host-restatement pin, no tensor2img in between)
    rng = np.random.RandomState(3)
    a = rng.rand(40, 33) * 255
    b = np.clip(a + rng.randn(40, 33) * 12, 0, 255)
    a3, b3 = np.stack([a, a * 0.5, 255 - a], -1), np.stack([b, b * 0.5, 255 - b], -1)
    res['raw_a'], res['raw_b'] = a, b
    res['raw_ssim'] = np.array(util.calculate_ssim(a, b))
    res['raw_ssim3'] = np.array(util.calculate_ssim(a3, b3))
    res['raw_same'] = np.array(util.calculate_ssim(a, a))
    print('  raw %.12f  raw3 %.12f  same %.12f  grey %.12f' % (res['raw_ssim'], res['raw_ssim3'], res['raw_same'], res['grey']))
    for bad in ((a, a[:-1]), (a3[..., :2], b3[..., :2])):
        try:
            r = util.calculate_ssim(*bad)
            print('  (reference returns %r for shapes %s)' % (r, bad[0].shape))
        except ValueError as e:
            print('  reference raises:', e)
    np.savez_compressed(os.path.join(OUT, 'ssim.npz'), **res)


def gen_imresize():
    """MATLAB-style bicubic imresize (codes/data/util.py:276-412) and augment (94-106): outputs of
    the imported reference on seeded inputs."""
    import random
    U = RI.data_util()
    res = {}
    cases = ((0, 37, 45, 0.25), (1, 64, 48, 0.25), (2, 33, 20, 0.5), (3, 9, 12, 2), (4, 7, 5, 4),
             (5, 31, 50, 1.0 / 3), (6, 40, 40, 0.7))
    for i, h, w, sc in cases:
        x = synth.image_batch(60 + i, 1, 3, h, w, name='imresize.x')[0]
        y = U.imresize(x.clone(), sc, True)
        ynp = U.imresize_np(x.permute(1, 2, 0).contiguous().numpy().copy(), sc, True)
        assert np.abs(ynp - y.permute(1, 2, 0).numpy()).max() < 1e-6
        res['scale%d' % i] = np.array(sc)
        res['shape%d' % i] = np.array([h, w])
        res['y%d' % i] = npy(y)
    x = synth.image_batch(70, 1, 3, 6, 8, name='imresize.x')[0]
    res['y_noaa'] = npy(U.imresize(x.clone(), 0.5, False))
    # augment: the 8 outcomes of (hflip, vflip, rot90) as the reference draws them from `random`
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(3, 4, 2)       # HWC
    outs, draws = [], []
    for seed in range(12):
        random.seed(seed)
        o = U.augment([a.copy()], True, True)[0]
        random.seed(seed)
        draws.append([random.random() < 0.5 for _ in range(3)])
        outs.append(np.ascontiguousarray(o).reshape(-1))
    res['aug_out'] = np.stack(outs)
    res['aug_draws'] = np.array(draws)
    np.savez_compressed(os.path.join(OUT, 'imresize.npz'), **res)


if __name__ == '__main__':
    assert RI.available(), 'reference tree not found — fixtures can only be generated in the build container'
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['rdb', 'rrdbnet_small', 'rrdbnet_full', 'disc', 'disc_variants', 'vgg', 'train_step',
                             'psnr', 'imresize', 'metrics', 'metrics_y', 'srresnet', 'rrdbnet_full_grad', 'disc_sn', 'train_steps3', 'rrdbnet_full_grad_fp16emu', 'disc_sn_two', 'rrdbnet_small_fp16emu',
                             'train_step_full', 'sr_infer', 'ssim']
    for w in which:
        print('[gen_golden]', w)
        globals()['gen_' + w]()
    print('done ->', OUT)
