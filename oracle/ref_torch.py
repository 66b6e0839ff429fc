"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the ESRGAN+ hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``esrganplus_amd``) never does; it fails loudly without the
HIP extension.

The reference expresses the path as stock ``torch.nn`` modules (fp32).  This file restates the
same arithmetic *functionally* over a plain state dict — no ``nn.Module`` tree — with the
Gaussian-noise draws made explicit (``z`` tensors passed in) so that results are reproducible
without torch's global generator.  Every function cites the reference lines it follows.
Parity of this restatement with the imported reference is pinned by ``oracle/gen_golden.py``
(run in the build container where ``/root/reference`` exists) and re-checked against the
committed fixtures in ``tests/test_oracle.py``.

Third-party arithmetic: ``torch.nn.functional.conv2d / batch_norm / linear / max_pool2d /
interpolate(nearest)`` on CPU (torch 2.10, oneDNN) — the same library calls the reference's
``nn.Conv2d`` etc. dispatch to.  An independent plain-C restatement of those primitives lives
in ``oracle/conv_ref.c`` and is cross-checked against this file in ``tests/test_oracle.py``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SIGMA = 0.1          # GaussianNoise sigma, block.py:111
NEG_SLOPE = 0.2      # LeakyReLU slope, block.py:12


def _conv(x, sd, key, stride=1):
    """conv_block's Conv2d with zero padding (k-1)//2 — block.py:125-142, 55-58."""
    w = sd[key + '.weight']
    b = sd.get(key + '.bias')
    pad = (w.shape[-1] - 1) // 2
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def _lrelu(x):
    return F.leaky_relu(x, NEG_SLOPE)


def gaussian_noise(x, z):
    """GaussianNoise.forward, block.py:117-122: x + z*(sigma*x); identity when z is None (eval)."""
    if z is None:
        return x
    return x + z * (SIGMA * x)


def rdb_forward(x, sd, p, z=None):
    """ResidualDenseBlock_5C.forward — block.py:260-268 (same in test_image/block.py:224-232)."""
    x1 = _lrelu(_conv(x, sd, p + '.conv1.0'))
    x2 = _lrelu(_conv(torch.cat((x, x1), 1), sd, p + '.conv2.0'))
    x2 = x2 + F.conv2d(x, sd[p + '.conv1x1.weight'])            # block.py:263, bias-free 1x1
    x3 = _lrelu(_conv(torch.cat((x, x1, x2), 1), sd, p + '.conv3.0'))
    x4 = _lrelu(_conv(torch.cat((x, x1, x2, x3), 1), sd, p + '.conv4.0'))
    x4 = x4 + x2                                                 # block.py:266
    x5 = _conv(torch.cat((x, x1, x2, x3, x4), 1), sd, p + '.conv5.0')
    return gaussian_noise(x5 * 0.2 + x, z)                       # block.py:268


def rrdb_forward(x, sd, p, zs=None, z_rrdb=None):
    """RRDB.forward — block.py:287-291; ``z_rrdb`` is the extra noise layer of the inference
    copy, test_image/block.py:250,256."""
    zs = zs or (None, None, None)
    out = rdb_forward(x, sd, p + '.RDB1', zs[0])
    out = rdb_forward(out, sd, p + '.RDB2', zs[1])
    out = rdb_forward(out, sd, p + '.RDB3', zs[2])
    return gaussian_noise(out * 0.2 + x, z_rrdb)


def noise_shapes(x_shape, nb, variant='codes'):
    """Shapes of the normal_() draws of one training forward, in module execution order
    (block.py:120 draws the full activation shape once per noise layer)."""
    b, _, h, w = x_shape
    per = 3 if variant == 'codes' else 4
    return [(b, 64, h, w)] * (per * nb)


def rrdbnet_forward(x, sd, nb=None, z=None, variant='codes'):
    """RRDBNet.forward — architecture.py:47-78 (RRDB_Net: test_image/architecture.py:7-38).

    ``z``: None (eval) or a flat list of noise tensors in execution order: per RRDB
    [RDB1, RDB2, RDB3] for the ``codes`` copy, [RDB1, RDB2, RDB3, RRDB] for ``test_image``.
    """
    if nb is None:
        nb = sum(1 for k in sd if k.endswith('.RDB1.conv1x1.weight'))
    per = 3 if variant == 'codes' else 4
    fea = _conv(x, sd, 'model.0')                                # fea_conv, architecture.py:55
    t = fea
    for i in range(nb):
        zi = None if z is None else z[per * i: per * i + per]
        t = rrdb_forward(t, sd, 'model.1.sub.%d' % i,
                         None if zi is None else zi[:3],
                         None if (zi is None or per == 3) else zi[3])
    t = _conv(t, sd, 'model.1.sub.%d' % nb)                      # LR_conv, architecture.py:58
    t = fea + t                                                  # ShortcutBlock, block.py:84-86
    for key in ('model.3', 'model.6'):                           # upconv_blcok, block.py:315-322
        t = F.interpolate(t, scale_factor=2, mode='nearest')
        t = _lrelu(_conv(t, sd, key))
    t = _lrelu(_conv(t, sd, 'model.8'))                          # HR_conv0, architecture.py:70
    return _conv(t, sd, 'model.10')                              # HR_conv1, architecture.py:71


D_LAYOUT = [(0, None, 1), (2, 3, 2), (5, 6, 1), (8, 9, 2), (11, 12, 1), (14, 15, 2),
            (17, 18, 1), (20, 21, 2), (23, 24, 1), (26, 27, 2)]   # (conv idx, bn idx, stride)


def discriminator_forward(x, sd, training=True, momentum=0.1, eps=1e-5, update_stats=True, size=128):
    """Discriminator_VGG_128.forward — architecture.py:87-129 (size 96 / 192: architecture.py:178-270, the same
    stack on a 3x3 final map / with one more 512-channel conv pair).  BatchNorm2d(affine) in train mode
    uses biased batch variance for normalisation and updates running stats with the unbiased
    one (torch.nn.BatchNorm2d semantics, block.py:28-32).  ``sd`` running stats are updated
    in place when ``training and update_stats`` (num_batches_tracked += 1 per call)."""
    t = x
    layout = D_LAYOUT + ([(29, 30, 1), (32, 33, 2)] if size == 192 else [])
    for ci, bi, stride in layout:
        t = _conv(t, sd, 'features.%d' % ci, stride)
        if bi is not None:
            p = 'features.%d' % bi
            rm, rv = sd[p + '.running_mean'], sd[p + '.running_var']
            if training and not update_stats:
                rm, rv = rm.clone(), rv.clone()
            t = F.batch_norm(t, rm, rv, sd[p + '.weight'], sd[p + '.bias'],
                             training, momentum, eps)
            if training and update_stats:
                sd[p + '.num_batches_tracked'] += 1
        t = _lrelu(t)
    t = t.reshape(t.size(0), -1)                                 # architecture.py:127 (C,H,W order)
    t = _lrelu(F.linear(t, sd['classifier.0.weight'], sd['classifier.0.bias']))
    return F.linear(t, sd['classifier.2.weight'], sd['classifier.2.bias'])


VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def vgg19_features_forward(x, sd, feature_layer=34, use_input_norm=True):
    """VGGFeatureExtractor.forward — architecture.py:279-307; body = torchvision vgg19 cfg 'E'
    ``features[:feature_layer+1]`` (conv5_4 before ReLU for 34)."""
    from collections import OrderedDict  # noqa: F401
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
           512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
    if use_input_norm:
        mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
        std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
        x = (x - mean) / std                                     # architecture.py:304-305
    idx = 0
    t = x
    for v in cfg:
        if idx > feature_layer:
            break
        if v == 'M':
            t = F.max_pool2d(t, 2, 2)
            idx += 1
        else:
            t = _conv(t, sd, 'features.%d' % idx)
            idx += 1
            if idx <= feature_layer:
                t = F.relu(t)
            idx += 1
    return t


# --------------------------------------------------------------------------------------------
# losses of the ESRGAN+ train step — SRRaGAN_model.py:113-168, loss.py:6-38
# --------------------------------------------------------------------------------------------

def bce_logits(x, target_is_real):
    """GANLoss('vanilla') = BCEWithLogitsLoss against a constant label — loss.py:13-14,30-38."""
    t = torch.ones_like(x) if target_is_real else torch.zeros_like(x)
    return F.binary_cross_entropy_with_logits(x, t)


def generator_losses(fake_H, var_H, fake_fea, real_fea, pred_g_fake, pred_d_real,
                     l_pix_w=0.01, l_fea_w=1.0, l_gan_w=0.005):
    """SRRaGAN_model.py:122-138 with train_ESRGANplus.json weights."""
    l_pix = l_pix_w * F.l1_loss(fake_H, var_H)
    l_fea = l_fea_w * F.l1_loss(fake_fea, real_fea)
    l_gan = l_gan_w * (bce_logits(pred_d_real - pred_g_fake.mean(), False) +
                       bce_logits(pred_g_fake - pred_d_real.mean(), True)) / 2
    return l_pix, l_fea, l_gan


def discriminator_losses(pred_d_real, pred_d_fake):
    """SRRaGAN_model.py:149-154."""
    l_real = bce_logits(pred_d_real - pred_d_fake.mean(), True)
    l_fake = bce_logits(pred_d_fake - pred_d_real.mean(), False)
    return l_real, l_fake


# --------------------------------------------------------------------------------------------
# metric restatement — codes/utils/util.py:71-95,107-114 and codes/train.py:143-148
# --------------------------------------------------------------------------------------------

def tensor2img(t):
    """util.tensor2img for a single 3-D (C,H,W) RGB tensor in [0,1] -> uint8 HWC, BGR order."""
    a = t.detach().squeeze().float().cpu().clamp(0, 1).numpy()
    a = np.transpose(a[[2, 1, 0], :, :], (1, 2, 0))
    return (a * 255.0).round().astype(np.uint8)


def calculate_psnr(img1, img2):
    """util.calculate_psnr — images in [0,255]."""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def psnr_sr(sr, hr, scale=4):
    """Validation PSNR as codes/train.py:131-148: tensor2img both, /255, crop `scale` px, *255."""
    a = tensor2img(sr) / 255.
    b = tensor2img(hr) / 255.
    a = a[scale:-scale, scale:-scale, :]
    b = b[scale:-scale, scale:-scale, :]
    return calculate_psnr(a * 255, b * 255)
