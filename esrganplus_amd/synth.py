"""Deterministic synthetic weights / inputs for parity tests and benchmarks.

There is no network (no pretrained ``nESRGANplus.pth``, reference README.md:26), so every
parity and bench run uses seeded synthetic state dicts.  The generator is numpy-only
(``np.random.Philox``) so that this container (where the golden fixtures are produced by
importing the reference) and the GPU box (where the reference does not exist) build
bit-identical tensors.

Key layouts follow the reference exactly:
  RRDBNet            codes/models/modules/architecture.py:47-78, block.py:232-291
  Discriminator      codes/models/modules/architecture.py:87-129
  VGG19 features     codes/models/modules/architecture.py:279-307 (torchvision cfg 'E')
"""
from collections import OrderedDict
import zlib

import numpy as np
import torch

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def _rng(seed, name):
    # one independent Philox stream per (seed, tensor name)
    key = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return np.random.Generator(np.random.Philox(key=[int(seed), key]))


def _uniform(seed, name, shape, bound):
    g = _rng(seed, name)
    a = g.random(size=shape, dtype=np.float64) * 2.0 - 1.0
    return torch.from_numpy((a * bound).astype(np.float32))


def _conv(sd, seed, key, cout, cin, k, bias=True, gain=1.0):
    """PyTorch-default-like init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (times ``gain``).

    SURVEY.md §8c: kaiming*0.1 (networks.py:104) drives a 23-block net's output to ~1e-4 and
    would make the 1e-3 gate vacuous, so parity weights use default-scale init.
    """
    fan_in = cin * k * k
    bound = gain / np.sqrt(fan_in)
    sd[key + '.weight'] = _uniform(seed, key + '.weight', (cout, cin, k, k), bound)
    if bias:
        sd[key + '.bias'] = _uniform(seed, key + '.bias', (cout,), 1.0 / np.sqrt(fan_in))


def rrdbnet_keys(nb):
    """(key-prefix, cout, cin, k, bias) for every conv of RRDBNet x4, in state-dict order."""
    out = [('model.0', 64, 3, 3, True)]
    for i in range(nb):
        for j in (1, 2, 3):
            p = 'model.1.sub.%d.RDB%d' % (i, j)
            out.append((p + '.conv1x1', 32, 64, 1, False))
            for k in range(1, 5):
                out.append((p + '.conv%d.0' % k, 32, 64 + 32 * (k - 1), 3, True))
            out.append((p + '.conv5.0', 64, 192, 3, True))
    out.append(('model.1.sub.%d' % nb, 64, 64, 3, True))
    out += [('model.3', 64, 64, 3, True), ('model.6', 64, 64, 3, True),
            ('model.8', 64, 64, 3, True), ('model.10', 3, 64, 3, True)]
    return out


def rrdbnet_state_dict(nb=23, seed=0, gain=1.0):
    sd = OrderedDict()
    for key, cout, cin, k, bias in rrdbnet_keys(nb):
        _conv(sd, seed, key, cout, cin, k, bias, gain)
    return sd


def srresnet_keys(nb, upsample_mode='pixelshuffle'):
    """(key-prefix, cout, cin, k) of SRResNet x4 (architecture.py:13-44 of the reference, networks.py:88-91)."""
    out = [('model.0', 64, 3, 3)]
    for i in range(nb):
        out += [('model.1.sub.%d.res.0' % i, 64, 64, 3), ('model.1.sub.%d.res.2' % i, 64, 64, 3)]
    out.append(('model.1.sub.%d' % nb, 64, 64, 3))
    if upsample_mode == 'pixelshuffle':       # [conv, PixelShuffle, ReLU] x 2, then HR_conv0, ReLU, HR_conv1
        out += [('model.2', 256, 64, 3), ('model.5', 256, 64, 3)]
    else:                                     # [Upsample, conv, ReLU] x 2
        out += [('model.3', 64, 64, 3), ('model.6', 64, 64, 3)]
    out += [('model.8', 64, 64, 3), ('model.10', 3, 64, 3)]
    return out


def srresnet_state_dict(nb=16, seed=0, upsample_mode='pixelshuffle'):
    sd = OrderedDict()
    for key, cout, cin, k in srresnet_keys(nb, upsample_mode):
        _conv(sd, seed, key, cout, cin, k, True)
    return sd


D_CONVS = [(0, 3, 64, 3), (2, 64, 64, 4), (5, 64, 128, 3), (8, 128, 128, 4), (11, 128, 256, 3),
           (14, 256, 256, 4), (17, 256, 512, 3), (20, 512, 512, 4), (23, 512, 512, 3),
           (26, 512, 512, 4)]
D_BNS = [(3, 64), (6, 128), (9, 128), (12, 256), (15, 256), (18, 512), (21, 512), (24, 512),
         (27, 512)]


def discriminator_layout(size=128):
    """(convs [(idx, cin, cout, ks)], bns [(idx, channels)], flattened features) of Discriminator_VGG_<size>
    (architecture.py:87-129 / 178-270): 96 = the 128 net on a 3x3 final map, 192 = one more 512-channel pair."""
    convs, bns = list(D_CONVS), list(D_BNS)
    if size == 192:
        convs += [(29, 512, 512, 3), (32, 512, 512, 4)]
        bns += [(30, 512), (33, 512)]
    return convs, bns, 512 * (16 if size == 128 else 9)


def discriminator_state_dict(seed=0, size=128):
    """Discriminator_VGG_128/96/192(in_nc=3, base_nf=64) — keys per SURVEY.md Appendix C."""
    sd = OrderedDict()
    convs, bns, nfeat = discriminator_layout(size)
    bn = dict(bns)
    for idx, cin, cout, k in convs:
        _conv(sd, seed, 'features.%d' % idx, cout, cin, k, True, gain=np.sqrt(3.0))
        if idx + 1 in bn:
            c = bn[idx + 1]
            p = 'features.%d' % (idx + 1)
            sd[p + '.weight'] = 1.0 + _uniform(seed, p + '.weight', (c,), 0.2)
            sd[p + '.bias'] = _uniform(seed, p + '.bias', (c,), 0.2)
            sd[p + '.running_mean'] = _uniform(seed, p + '.running_mean', (c,), 0.1)
            sd[p + '.running_var'] = 1.0 + _uniform(seed, p + '.running_var', (c,), 0.3)
            sd[p + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    for name, cout, cin in (('classifier.0', 100, nfeat), ('classifier.2', 1, 100)):
        b = 1.0 / np.sqrt(cin)
        sd[name + '.weight'] = _uniform(seed, name + '.weight', (cout, cin), b)
        sd[name + '.bias'] = _uniform(seed, name + '.bias', (cout,), b)
    return sd


def discriminator_sn_state_dict(seed=0):
    """Discriminator_VGG_128_SN (architecture.py:131-175 + spectral_norm.py:55-75): per layer ``weight_orig``, ``bias``,
    ``weight`` (buffer) and ``weight_u`` (buffer, unit norm)."""
    sd = OrderedDict()
    shapes, cin = [], 3
    for w in (64, 128, 256, 512, 512):
        shapes += [(w, cin, 3, 3), (w, w, 4, 4)]
        cin = w
    layers = [('conv%d' % i, sh) for i, sh in enumerate(shapes)] + [('linear0', (100, 512 * 16)), ('linear1', (1, 100))]
    for name, sh in layers:
        fan_in = int(np.prod(sh[1:]))
        w = _uniform(seed, name + '.weight_orig', sh, np.sqrt(3.0) / np.sqrt(fan_in))
        sd[name + '.weight_orig'] = w
        sd[name + '.bias'] = _uniform(seed, name + '.bias', (sh[0],), 1.0 / np.sqrt(fan_in))
        sd[name + '.weight'] = w.clone()
        u = normal_like(seed, name + '.weight_u', (sh[0],))
        sd[name + '.weight_u'] = u / u.norm()
    return sd


def vgg19_conv_indices(feature_layer=34):
    idx, out, cin = 0, [], 3
    for v in VGG19_CFG:
        if v == 'M':
            idx += 1
        else:
            if idx <= feature_layer:
                out.append((idx, cin, v))
            cin = v
            idx += 2
    return out


def vgg19_state_dict(seed=0, feature_layer=34):
    """He-scaled synthetic VGG19 ``features[:35]`` weights (ImageNet weights are unobtainable:
    SURVEY.md §8c — default-init weights give a ~7e-7 feature loss, numerically useless)."""
    sd = OrderedDict()
    for idx, cin, cout in vgg19_conv_indices(feature_layer):
        _conv(sd, seed, 'features.%d' % idx, cout, cin, 3, True, gain=np.sqrt(6.0))
    return sd


def image_batch(seed, b, c, h, w, name='x'):
    """U[0,1) image-like input, NCHW float32."""
    g = _rng(seed, name)
    return torch.from_numpy(g.random(size=(b, c, h, w), dtype=np.float32))


def normal_like(seed, name, shape):
    g = _rng(seed, name)
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float32))
